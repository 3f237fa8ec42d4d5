"""Moves a known number of bytes at each access width (test hook sg_test_pmc_calibrate) so that rocprofv3's FETCH_SIZE /
WRITE_SIZE can be calibrated for this library's access patterns (MI355X_MICROARCH.md section HBM prescribes exactly this
for widths other than the 16-byte-per-lane stream).  Run under the profiler, once per counter:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/c_f -o f --output-format rocpd -- python tools/pmc_calibrate.py 1024
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/c_w -o w --output-format rocpd -- python tools/pmc_calibrate.py 1024
    python tools/rocpd_pmc.py <db>      # KB per dispatch of k_calib_read<W> / k_calib_write<W>: divide by MBYTES*1024

tools/profile_calibration.sh does all of it and writes profiles/<tag>_pmc_calibration.json."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simgan_amd import _lib  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = _lib.Context.default()
fn = _lib.load_test().sg_test_pmc_calibrate
_lib.check_test(fn(ctx.h, mb))
print(f"moved {mb} MiB per kernel")
