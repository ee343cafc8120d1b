#!/usr/bin/env python3
"""Compare two .npy dumps (tools/fwd_dump.py): max-abs / mean-abs difference per file."""
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
d = np.abs(a.astype(np.float64) - b)
print(sys.argv[1], sys.argv[2], "max-abs", d.max(), "mean-abs", d.mean(), "ref absmax", np.abs(b).max(), "finite", np.isfinite(a).all())
