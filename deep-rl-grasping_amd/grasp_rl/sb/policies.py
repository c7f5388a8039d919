"""Policy *selectors*.  In stable-baselines these classes build TF graphs; the reference only passes
the class objects to the model constructors (sb_helper.py:90,93,96,159,210) together with
``policy_kwargs`` (``layers``, ``layer_norm``, ``cnn_extractor``).  Here they carry the choice to the
HIP engine: which feature extractor, whether layer-norm was requested."""


class BasePolicy:
    feature_extraction = "mlp"
    layer_norm = False
    family = "common"


class SacMlpPolicy(BasePolicy):
    family = "sac"


class SacCnnPolicy(BasePolicy):
    family = "sac"
    feature_extraction = "cnn"


class SacLnMlpPolicy(SacMlpPolicy):
    layer_norm = True


class SacLnCnnPolicy(SacCnnPolicy):
    layer_norm = True


class DqnMlpPolicy(BasePolicy):
    family = "deepq"


class DqnLnMlpPolicy(DqnMlpPolicy):
    layer_norm = True


class DqnCnnPolicy(BasePolicy):
    family = "deepq"
    feature_extraction = "cnn"


class BdqMlpActPolicy(BasePolicy):
    family = "bdq"


class ActorCriticMlpPolicy(BasePolicy):
    family = "common"


class ActorCriticCnnPolicy(BasePolicy):
    family = "common"
    feature_extraction = "cnn"


class AugmentedNatureCnn:
    """Picklable stand-in for the reference's ``create_augmented_nature_cnn(n)`` closure
    (custom_obs_policy.py:6-44) in ``policy_kwargs['cnn_extractor']``: carries the name the engine recognises and
    ``num_direct_features``.  Used when a zip is loaded whose cloudpickled closure is not (or may not be)
    unpickled: the extractor is then inferred from the parameter names / shapes."""

    def __init__(self, num_direct_features=1):
        self.__name__ = "augmented_nature_cnn"
        self.num_direct_features = int(num_direct_features)

    def __call__(self, *a, **k):
        raise NotImplementedError("the feature extractor runs in the HIP engine; this object only selects it")


def infer_sac_policy_kwargs(params):
    """policy class + ``policy_kwargs`` of a SAC zip from its TF variable names / shapes:
    ``model/pi/cnn1`` -> augmented extractor with fc0 rows - 512 direct features, ``model/pi/c1`` -> nature_cnn,
    neither -> MLP; ``layers`` = widths of ``model/pi/fc<l>/kernel:0``."""
    layers, l = [], 0
    while "model/pi/fc%d/kernel:0" % l in params:
        layers.append(int(params["model/pi/fc%d/kernel:0" % l].shape[1]))
        l += 1
    kw = {"layers": layers}
    if any(n.startswith("model/pi/cnn1/") for n in params):
        kw["cnn_extractor"] = AugmentedNatureCnn(int(params["model/pi/fc0/kernel:0"].shape[0]) - 512)
        return SacCnnPolicy, kw
    if any(n.startswith("model/pi/c1/") for n in params):
        return SacCnnPolicy, kw
    return SacMlpPolicy, kw


def extractor_from_kwargs(policy, policy_kwargs):
    """('mlp' | 'nature' | 'augmented', n_direct).  The reference's ``cnn_extractor`` is a TF closure
    (custom_obs_policy.py:6-44); it is recognised by name and its ``num_direct_features`` is read
    from the closure -- it is never called."""
    if getattr(policy, "feature_extraction", "mlp") != "cnn":
        return "mlp", 0
    fn = (policy_kwargs or {}).get("cnn_extractor")
    if fn is None:
        return "nature", 0
    name = getattr(fn, "__name__", "")
    if name != "augmented_nature_cnn":
        raise NotImplementedError("cnn_extractor %r: only the default nature_cnn and the reference's "
                                  "augmented_nature_cnn are implemented as HIP kernels" % name)
    n_direct = getattr(fn, "num_direct_features", None)
    if n_direct is None and getattr(fn, "__closure__", None):
        ints = [c.cell_contents for c in fn.__closure__ if isinstance(c.cell_contents, int)]
        n_direct = ints[0] if ints else None
    if n_direct is None:
        raise ValueError("cannot determine num_direct_features of the augmented_nature_cnn closure")
    return "augmented", int(n_direct)
