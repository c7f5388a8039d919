"""`stable_baselines.results_plotter.load_results / ts2xy` (call site: /root/reference/scripts/plot.py:7,64,76):
read the monitor CSVs a training run leaves behind -- ours (grasp_rl.sb.monitor) and the ones the reference
ships under trained_models/*/log_file.monitor.csv (its fork adds the columns s, c, timesteps)."""
import glob
import json
import os

import numpy as np
import pandas

X_TIMESTEPS, X_EPISODES, X_WALLTIME = "timesteps", "episodes", "walltime_hrs"
POSSIBLE_X_AXES = [X_TIMESTEPS, X_EPISODES, X_WALLTIME]


class LoadMonitorResultsError(Exception):
    pass


def get_monitor_files(path):
    return sorted(glob.glob(os.path.join(path, "*monitor.csv")))


def load_results(path):
    """All `*monitor.csv` files of a folder as one DataFrame sorted by wall-clock time (column `t` is made
    relative to the earliest `t_start`); extra columns are kept."""
    files = get_monitor_files(path)
    if not files:
        raise LoadMonitorResultsError("no monitor files of the form *monitor.csv found in %s" % path)
    frames, headers = [], []
    for name in files:
        with open(name, "rt") as f:
            first = f.readline()
            if not first.startswith("#"):
                raise LoadMonitorResultsError("%s does not start with the #{json} header line" % name)
            header = json.loads(first[1:])
            frame = pandas.read_csv(f, index_col=None)
        headers.append(header)
        frame["t"] += header["t_start"]
        frames.append(frame)
    df = pandas.concat(frames)
    df.sort_values("t", inplace=True)
    df.reset_index(inplace=True, drop=True)
    df["t"] -= min(h["t_start"] for h in headers)
    return df


def ts2xy(timesteps, xaxis, y_column="r"):
    """x / y arrays of a learning curve: episode return (or another column) against cumulated timesteps,
    episode number, or wall-clock hours."""
    if xaxis == X_TIMESTEPS:
        x = np.cumsum(timesteps.l.values)
    elif xaxis == X_EPISODES:
        x = np.arange(len(timesteps))
    elif xaxis == X_WALLTIME:
        x = timesteps.t.values / 3600.0
    else:
        raise NotImplementedError(xaxis)
    return x, timesteps[y_column].values
