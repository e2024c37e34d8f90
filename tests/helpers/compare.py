"""Comparison helpers shared by the GPU parity tests."""
import json
import os

import numpy as np

RECORD = []        # (label, entries, outside strict, max err / scale) of every grad_close call of the session


def grad_close(got, ref, strict=2e-4, loose=2e-2, frac=0.0015, exact=False, max_outside=None, label=None):
    """Gradient comparison for a piecewise-smooth loss (see the header of test_ssw_gpu.py): every entry within
    `loose` of the largest reference entry, and at most max(frac * size, 8) entries outside `strict` of it
    (one swapped pair of points = 6 entries).  Round 2 measured the counts of the whole suite on MI355X
    (gpurun_out/grad_close_counts.json: 0 for most cases, worst 32 of 24576 = 0.13 % at n = 8192) and lowered the
    allowance from 0.5 % / 12 entries to 0.15 % / 8.  `exact=True`: every entry inside `strict`.
    `max_outside` pins the COUNT of entries outside the strict bound for a fixture case: a
    regression that breaks a fraction of the entries below the allowance is still caught.  Returns that count."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = np.abs(ref).max()
    err = np.abs(got - ref)
    outside = int((err > strict * scale).sum())
    RECORD.append({"label": label or os.environ.get("PYTEST_CURRENT_TEST", "?"), "entries": int(err.size),
                   "outside_strict": outside, "max_err_over_scale": float(err.max() / max(scale, 1e-300))})
    if exact:
        assert err.max() < strict * scale, (err.max(), scale)
        return outside
    assert err.max() < loose * scale, (err.max(), scale)
    allowed = max(frac * err.size, 8)
    assert outside <= allowed, (outside, err.size, err.max(), scale)
    if max_outside is not None:
        assert outside <= max_outside, f"{outside} entries outside the strict bound, pinned at <= {max_outside}"
    return outside


def dump_record(path):
    if RECORD:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(RECORD, fh, indent=1)
