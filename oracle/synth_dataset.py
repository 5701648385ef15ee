"""ORACLE helper (test infrastructure): a small synthetic on-disk dataset in the reference's
two-person format (codes/datasets/mul_dataset.py:73-127): `<motion_dir>/<name>.npy` of shape
(2, frames + 1, 263) float32 whose LAST row is the init pose, `<text_dir>/<name>.txt` with lines
`caption1_caption2#tok tok#0.0#0.0`, a split file with one name per line, and mean / std of
shape (263 + 4,).  Deterministic (crc32-seeded by name), so the golden generator (which feeds it to
the reference's own dataset class) and the tests (here and on the GPU box) see identical bytes.
"""
import json
import os
import types

import numpy as np

from . import fill

F = 263
# name -> number of motion frames (rows = frames + 1): below 20 and from 200 up are dropped by the
# reference (:83), below 90 exercises the padding branch (:187-192), above 90 the random-shift branch
SEQS = [("S001", 150), ("S002", 60), ("S003", 12), ("S004", 91), ("S005", 230), ("S006", 90), ("S007", 199),
        ("S008", 20), ("S009", 120), ("S010", 33)]
CAPTIONS = [["one pushes the other", "one is pushed by the other"], ["two people hug"],
            ["a person hands a cup over", "a person receives a cup"], ["they shake hands"]]


def write(root):
    """Creates the files under `root`; returns the `opt` namespace the dataset classes expect."""
    mdir, tdir, meta = (os.path.join(root, d) for d in ("new_joint_vecs", "texts", "meta"))
    for d in (mdir, tdir, meta):
        os.makedirs(d, exist_ok=True)
    labels = {}
    for k, (name, frames) in enumerate(SEQS):
        m = (fill.tensor_for("synth.motion." + name, (2, frames + 1, F)) * 10.0).numpy().astype(np.float32)
        np.save(os.path.join(mdir, name + ".npy"), m)
        lines = []
        for j in range(1 + k % 3):
            caps = CAPTIONS[(k + j) % len(CAPTIONS)]
            lines.append("%s#%s#0.0#0.0" % ("_".join(caps), " ".join(w + "/NOUN" for w in caps[0].split())))
        with open(os.path.join(tdir, name + ".txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
        labels[name] = k % 2
    with open(os.path.join(root, "train.txt"), "w") as f:
        f.write("\n".join([n for n, _ in SEQS] + ["MISSING"]) + "\n")       # a missing file is skipped (:125-127)
    with open(os.path.join(root, "labels.json"), "w") as f:
        json.dump(labels, f)
    return types.SimpleNamespace(
        dataset_name="ntu_mul", joints_num=22, motion_dir=mdir, text_dir=tdir, meta_dir=meta, is_train=True,
        feat_bias=5.0, limit_data_num=-1, cap_id=False, cap_same=False, max_text_len=20, batch_size=4,
        split_file=os.path.join(root, "train.txt"), label_path=os.path.join(root, "labels.json"))


def stats(dtype=np.float32):
    mean = (fill.tensor_for("synth.mean", (F + 4,)) * 10.0).numpy().astype(dtype)
    std = (1.0 + 0.5 * np.abs((fill.tensor_for("synth.std", (F + 4,)) * 10.0).numpy())).astype(dtype)
    return mean, std
