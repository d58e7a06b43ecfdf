"""The kernels' gfx950 listing is free of the miscompile that gave round 2's four-wave build its wrong gradients
(tools/probe/exec_prologue_check.py): no vector instruction at the head of a loop-exit block ahead of the exec restore.
The checker itself is pinned on the two shapes it has to tell apart."""
import shutil
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools" / "probe"))
import exec_prologue_check as chk  # noqa: E402

BAD = """
_Z6kernelv:
.LBB0_250:
\tds_write_b64 v4, v[2:3]
\ts_andn2_b64 exec, exec, s[16:17]
\ts_cbranch_execnz .LBB0_250
.LBB0_251:                              ;   in Loop: Header=BB0_6 Depth=1
\tv_mov_b32_e32 v23, v51
\ts_movk_i32 s86, 0xff78
\ts_or_b64 exec, exec, s[14:15]
\ts_endpgm
"""

GOOD = """
_Z6kernelv:
.LBB0_250:
\tds_write_b64 v4, v[2:3]
\ts_andn2_b64 exec, exec, s[16:17]
\ts_cbranch_execnz .LBB0_250
.LBB0_251:
\ts_or_b64 exec, exec, s[14:15]
\tv_mov_b32_e32 v23, v51
\ts_cbranch_execz .LBB0_253
.LBB0_252:
\tv_add_f64 v[0:1], v[0:1], v[0:1]
\ts_or_b64 exec, exec, s[20:21]
.LBB0_253:
\ts_endpgm
"""


def test_checker_tells_the_two_shapes_apart(tmp_path):
    bad, good = tmp_path / "bad.s", tmp_path / "good.s"
    bad.write_text(BAD)
    good.write_text(GOOD)
    hits = chk.scan(str(bad))
    assert [(h[1], h[3], h[5]) for h in hits] == [(".LBB0_251", "v_mov_b32_e32 v23, v51", True)]
    # program code at the tail of an `if` body ahead of an inner region's exec restore is listed but is not a finding
    assert [h[5] for h in chk.scan(str(good))] == [False]


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc to make the listing")
def test_product_kernels_hold_no_copy_ahead_of_an_exec_restore():
    import __graft_entry__ as ge

    listing = ge.device_listing()
    text = listing.read_text()
    assert text.count(".amdhsa_kernel ") >= 40          # every kernel of the library is in the listing
    dead = [h for h in chk.scan(str(listing)) if h[5]]
    assert not dead, dead[:3]
