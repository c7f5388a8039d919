"""Launcher-side helper for data-parallel training with scripts that were written for one process, e.g. the reference's

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m grasp_rl.dp_run \\
        manipulation_main/training/train_stable_baselines.py train --config config/gripper_grasp.yaml --algo SAC \\
        --model_dir trained_models/SAC_dp

Every rank runs the script unchanged (``runpy``, ``__name__ == "__main__"``).  What a one-process script cannot know is
done here: ``GRL_DATA_PARALLEL=auto`` switches ``grasp_rl.sb.SAC`` to one replica of WORLD_SIZE (module docstring of
grasp_rl/sb/sac.py); directory arguments the script CREATES (``--model_dir``, train_stable_baselines.py:27-29
``os.mkdir(args.model_dir)``) get a per-rank suffix on ranks > 0 -- rank 0 keeps the name the user gave, so its model
zips, ``vecnormalize.pkl``, evaluation logs and Monitor file land where a single-process run puts them, while the other
ranks' Monitor files (their environments' episodes) go to ``<dir>.rank<k>``."""
import os
import runpy
import sys

PER_RANK_DIR_OPTIONS = ("--model_dir",)


def rewrite_argv(argv, rank):
    out = list(argv)
    if rank == 0:
        return out
    for k, a in enumerate(out):
        for opt in PER_RANK_DIR_OPTIONS:
            if a == opt and k + 1 < len(out):
                out[k + 1] = out[k + 1].rstrip("/") + ".rank%d" % rank
            elif a.startswith(opt + "="):
                out[k] = a.rstrip("/") + ".rank%d" % rank
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m grasp_rl.dp_run <script.py> [script arguments ...]")
    os.environ.setdefault("GRL_DATA_PARALLEL", "auto")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for the exchange buffers (before HIP initialises)
    rank = int(os.environ.get("RANK", "0"))
    script = argv[0]
    sys.argv = [script] + rewrite_argv(argv[1:], rank)
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
