"""GPU diagnostic: latency of a device-scope release/acquire flag between workgroups of one launch."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from simgan_amd import _lib  # noqa: E402

ctx = _lib.Context.default()
fn = _lib.load_test().sg_test_flag_probe
for mode, np_, nc, words in ((0, 96, 100, 2048), (1, 96, 100, 2048), (2, 96, 100, 2048), (2, 96, 100, 8192), (2, 32, 100, 2048), (0, 1, 8, 256), (1, 1, 8, 256), (2, 1, 8, 256),
                             (3, 1, 8, 256), (3, 16, 16, 2048), (3, 16, 16, 8192), (3, 24, 8, 8192), (2, 16, 16, 2048), (2, 16, 16, 8192)):
    st = (C.c_longlong * (2 * (np_ + nc)))()
    sums = (C.c_float * nc)()
    _lib.check_test(fn(ctx.h, mode, np_, nc, words, st, sums))
    s = np.array(st, dtype=np.int64).reshape(-1, 2)
    p, c = s[:np_], s[np_:]
    expect = words * sum(range(1, np_ + 1))
    good = all(abs(x - expect) < 1e-3 * expect for x in sums)
    print(f"mode {mode} ({'fence' if mode == 0 else 'one XCD, plain stores + sc1 loads' if mode == 3 else 'write-through'}): producers {np_} x {words * 4 // 1024} KiB, consumers {nc}: data ok {good}; release fence+atomic {10 * (p[:, 1] - p[:, 0]).max()} ns max "
          f"(median {10 * int(np.median(p[:, 1] - p[:, 0]))}); last producer pre-fence -> consumers saw flag {10 * (c[:, 0].min() - p[:, 0].max())}.."
          f"{10 * (c[:, 0].max() - p[:, 0].max())} ns; consumers read-back {10 * int(np.median(c[:, 1] - c[:, 0]))} ns median")
