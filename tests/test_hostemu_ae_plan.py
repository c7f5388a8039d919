"""Auto-encoder training launch plan (padded-conv tables incl. backward / weight-gradient masks, upsampling,
MSE, Keras-Adam) against oracle/autoencoder.py on CPU through the TEST-ONLY g++ emulation build."""
import ae_parity_util as au
from hostemu_backend import NumpyHostBackend


def test_ae_training_plan_matches_oracle(hostemu_lib):
    au.ae_check(backend=NumpyHostBackend(), lib_path=hostemu_lib, B=2, n_steps=2)
