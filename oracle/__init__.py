"""CPU oracle for the SAC / DQN / BDQ update hot path of BarisYazici/deep-rl-grasping.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the CPU baseline -- never as the thing that is
measured or shipped.  The product path (``deep-rl-grasping_amd/grasp_rl``) calls the HIP
library through its C ABI and fails loudly when that library is missing.

What it restates
----------------
The arithmetic of the reference's training hot path lives in third-party packages that are
NOT vendored under /root/reference and are not installable here:
``stable-baselines==2.10.1`` (reference ``setup.py:7``), ``tensorflow==1.14.0``
(``setup.py:8``), ``keras==2.2.4`` (``setup.py:12``) and the un-pinned ``bdq_sb`` submodule
(``.gitmodules:1-3``).  The oracle restates their published algorithms (SURVEY.md
Appendix A) anchored on the reference's own call sites:

* ``manipulation_main/training/custom_obs_policy.py:15-43``   (augmented Nature-CNN)
* ``manipulation_main/training/sb_helper.py:85-128``          (SAC wiring, policy kwargs)
* ``manipulation_main/gripperEnv/robot.py:183-228``           (observation layout)
* ``manipulation_main/gripperEnv/encoders.py:70-136``         (Keras depth auto-encoder)
* ``manipulation_main/gripperEnv/sensor.py:206-222``          (auto-encoder call site)

PARITY PINNING STATUS
---------------------
The reference's own tests (``tests_gripper/test_sim.py``) pin NOTHING on this path and the
reference implementation cannot be executed here, so there are no input->output golden
vectors from the reference itself: **parity is unpinned by reference tests**.  The oracle
is pinned only by fixture-derived *known-relationship* checks against artefacts the
reference ships: the saved SAC parameter set must satisfy the
soft-value relation V(s) ~= min Q(s, pi(s)) with the restated head wiring, and the shipped
Keras auto-encoder weights must reconstruct the real depth frames preserved in the
``vecnormalize.pkl`` files only with the restated padding / flatten conventions.
``scripts/make_golden.py`` runs those checks against /root/reference and commits the small
extracted fixtures + oracle outputs under ``tests/golden/``.
"""
