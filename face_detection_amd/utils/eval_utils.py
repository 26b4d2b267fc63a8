"""WiderFace result files from the batched detection buffers (reference evaluation.py:67-79).

The integer rows come from the GPU (face_detection_amd.widerface_rows_batch); formatting and
writing the text is host work."""
import os

import numpy as np


def format_widerface_result(name, xywh, scores, n=None):
    """Text of one result file: name, count, then `x y w h score` rows (evaluation.py:70-78)."""
    xywh = np.asarray(xywh)
    scores = np.asarray(scores)
    n = len(xywh) if n is None else int(n)
    lines = [name + "\n", "%d\n" % n]
    for box, score in zip(xywh[:n], scores[:n]):
        lines.append("%d %d %d %d %.3f\n" % (box[0], box[1], box[2], box[3], score))
    return "".join(lines)


def write_widerface_results(out_dir, names, xywh, scores, count):
    """One `<name>.txt` per image under out_dir (created if missing); returns the paths."""
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for b, name in enumerate(names):
        path = os.path.join(out_dir, name + ".txt")
        with open(path, "w") as f:
            f.write(format_widerface_result(name, xywh[b], scores[b], count[b]))
        paths.append(path)
    return paths
