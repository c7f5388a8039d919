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
