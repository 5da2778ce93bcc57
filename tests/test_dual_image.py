"""CPU test of the dual-use LDS image of csrc/fa_bwd_w64.hpp (`DualImg`): tools/dual_image.py restates the kernel's address
arithmetic (LDS-DMA source swizzle, row read, transposed column read) and checks, for E = 64 and 128, that every MFMA operand
read delivers the element the operand map asks for and that no read has a bank conflict under the LDS rules of
MI355X_MICROARCH.md (the property the backward kernels' single image per streamed tile rests on)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import dual_image  # noqa: E402


@pytest.mark.parametrize("E", [64, 128, 256])
def test_dual_image_reads_are_correct_and_conflict_free(E):
    assert dual_image.check(E) == {"row": 0, "col": 0}


def test_the_row_swizzles_of_the_forward_images_would_conflict_on_the_transposed_read():
    """why DualImg has its own swizzle: with RowImg's `row & 15` (E = 128) / `(row >> 1) & 7` (E = 64) the four rows of a
    transposed read share 64-byte bank groups (4-way / 2-way conflicts per 32-lane half); with DualImg's there is none"""
    assert dual_image.col_conflicts_with(128, lambda row: row & 15) >= 2 * 3
    assert dual_image.col_conflicts_with(64, lambda row: (row >> 1) & 7) >= 2 * 1
    for E in (64, 128, 256):
        assert dual_image.col_conflicts_with(E, lambda row, E=E: dual_image.xor_of(E, row)) == 0
