"""Tiny key/value logger with the ``stable_baselines.logger`` entry points the reference touches
(``from stable_baselines import logger`` in base_callbacks.py:13) and the SAC learn loop uses."""
import csv
import os
import sys

_kvs = {}
_dir = None
_csv_file, _csv_writer, _csv_keys = None, None, None


def configure(folder=None, format_strs=None):
    global _dir, _csv_file, _csv_writer, _csv_keys
    _dir = folder
    if folder is not None:
        os.makedirs(folder, exist_ok=True)
        _csv_file = open(os.path.join(folder, "progress.csv"), "wt")
        _csv_writer, _csv_keys = None, None


def get_dir():
    return _dir


def logkv(key, val):
    _kvs[key] = val


record = logkv


def logkvs(d):
    _kvs.update(d)


def getkvs():
    return dict(_kvs)


def dumpkvs(to_stdout=True):
    global _csv_writer, _csv_keys
    if not _kvs:
        return
    if to_stdout:
        w = max(len(k) for k in _kvs)
        sys.stdout.write("-" * (w + 20) + "\n")
        for k in sorted(_kvs):
            v = _kvs[k]
            sys.stdout.write("| %-*s | %-14s |\n" % (w, k, ("%.6g" % v) if isinstance(v, float) else str(v)))
        sys.stdout.write("-" * (w + 20) + "\n")
        sys.stdout.flush()
    if _csv_file is not None:
        if _csv_writer is None:
            _csv_keys = sorted(_kvs)
            _csv_writer = csv.DictWriter(_csv_file, fieldnames=_csv_keys, extrasaction="ignore")
            _csv_writer.writeheader()
        _csv_writer.writerow({k: _kvs.get(k, "") for k in _csv_keys})
        _csv_file.flush()
    _kvs.clear()


dump = dumpkvs


def info(*args):
    print(*args)


warn = error = debug = log = info


def warn(*args):
    import sys
    print("WARNING:", *args, file=sys.stderr)


class SummaryWriter:
    """Stand-in for the ``tf.summary.FileWriter`` stable-baselines hands to callbacks as ``locals['writer']``
    when ``tensorboard_log`` is set (the reference's TensorboardCallback calls
    ``self.locals['writer'].add_summary(tf.Summary(value=[tf.Summary.Value(tag=, simple_value=)]), step)``,
    /root/reference/manipulation_main/training/sb_helper.py:50-52; `tensorboard_logs` is set in
    config/full_depth_obs.yaml:65 and simplified_object_picking.yaml:74).  TensorBoard event files are out of
    scope: scalars that can be read off the summary object are appended to ``<logdir>/scalars.csv``
    (step, tag, value); anything else is accepted and dropped."""

    def __init__(self, log_dir, tb_log_name="run", new_tb_log=True):
        n = 1
        if os.path.isdir(log_dir):
            runs = [d for d in os.listdir(log_dir) if d.startswith(tb_log_name + "_") and d.rsplit("_", 1)[1].isdigit()]
            n = max([int(d.rsplit("_", 1)[1]) for d in runs], default=0) + (1 if new_tb_log or not runs else 0)
        self.log_dir = os.path.join(log_dir, "%s_%d" % (tb_log_name, n))
        os.makedirs(self.log_dir, exist_ok=True)
        self._f = open(os.path.join(self.log_dir, "scalars.csv"), "at")
        if self._f.tell() == 0:
            self._f.write("step,tag,value\n")
        self.n_summaries = 0

    @staticmethod
    def _scalars(summary):
        vals = getattr(summary, "value", None)
        if vals is None and isinstance(summary, dict):
            vals = summary.get("value")
        out = []
        for v in vals or []:
            tag = v.get("tag") if isinstance(v, dict) else getattr(v, "tag", None)
            val = v.get("simple_value") if isinstance(v, dict) else getattr(v, "simple_value", None)
            if tag is not None and val is not None:
                out.append((str(tag), float(val)))
        return out

    def add_summary(self, summary, global_step=None):
        self.n_summaries += 1
        for tag, val in self._scalars(summary):
            self._f.write("%s,%s,%.9g\n" % ("" if global_step is None else int(global_step), tag, val))

    def add_run_metadata(self, *a, **k):
        pass

    add_graph = add_event = add_run_metadata

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.close()
