"""Can ONE lane's write-through store of 16 (or 8) bytes be seen half-written from another XCD?  (csrc/sg_test.hip k_tear_probe)
Decides whether a hand-off can carry {values, step tag} in one 16-byte word.  GPU box:
    python tools/tear_probe.py [iterations per writer lane] [workgroup quads]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simgan_amd import _lib  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
quads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = _lib.Context.default()
t = _lib.load_test()
names = {0: "16-byte word, 16-byte aligned", 1: "16-byte word at an 8-byte offset", 2: "8-byte word, 8-byte aligned", 3: "8-byte word at a 4-byte offset",
         4: "8-byte word across a 64-byte boundary", 5: "16-byte word across a 64-byte boundary", 6: "16-byte word across a 128-byte line",
         7: "8-byte word across a 128-byte line"}
res = {}
for mode in (2, 3, 0, 1, 4, 5, 7, 6):
    tot = np.zeros(4, np.int64)
    for rep in range(3):
        out = np.zeros(4, np.int64)
        _lib.check_test(t.sg_test_tear_probe(ctx.h, mode, quads, iters, out.ctypes.data_as(C.POINTER(C.c_longlong))))
        tot[:2] += out[:2]; tot[2] = max(tot[2], out[2]); tot[3] += out[3]
    res[names[mode]] = {"torn_words": int(tot[0]), "words_read": int(tot[1]), "largest_value_seen": int(tot[2]), "reader_lanes_that_saw_a_change": int(tot[3])}
    print(names[mode], res[names[mode]], flush=True)
print(json.dumps({"iters_per_writer_lane": iters, "writer_workgroups": 4 * quads, "reader_workgroups": 4 * quads, "results": res}))
