"""Register and scratch budgets of the kernels, read from the BUILT library (the code objects' metadata notes), no GPU needed.

Several camera streams share a GPU, and which kernels' waves fit on a SIMD beside which is part of the design (DESIGN.md
section 4): four 21 x 21 PyrLK waves take 4 x 104 of a SIMD's 512 registers, and the short kernels of the other streams (merge pass:
two waves per SIMD; pyramid) must fit in the 96 that are left, or they wait for PyrLK waves to drain -- the merge kernel at 52
registers instead of 48 cost the 32-stream bench 17 % (profiles/r03_kernel_experiments.txt).  The register allocator's result
moves with any edit nearby, so the budgets are asserted here."""
import os
import struct
import subprocess

import msgpack
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "visualslam_amd", "lib", "libvisualslam_amd.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernels():
    """{demangled kernel name: metadata} of every gfx950 code object bundled in the library"""
    data = open(LIB, "rb").read()
    out, pos = {}, 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            break
        pos = i + len(MAGIC)
        n, = struct.unpack_from("<Q", data, i + 24)
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" not in triple or size == 0:
                continue
            elf = data[i + off:i + off + size]
            assert elf[:4] == b"\x7fELF"
            shoff, = struct.unpack_from("<Q", elf, 0x28)
            shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
            for s in range(shnum):
                sh = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + s * shentsize)
                if sh[1] != 7:        # SHT_NOTE
                    continue
                q, end = sh[4], sh[4] + sh[5]
                while q < end:
                    namesz, descsz, typ = struct.unpack_from("<III", elf, q)
                    q += 12
                    name = elf[q:q + namesz]
                    q += (namesz + 3) & ~3
                    desc = elf[q:q + descsz]
                    q += (descsz + 3) & ~3
                    if typ == 32 and name.startswith(b"AMDGPU"):      # NT_AMDGPU_METADATA
                        for k in msgpack.unpackb(desc, raw=False, strict_map_key=False)["amdhsa.kernels"]:
                            out[k[".name"]] = k
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    short = lambda d: d.replace("vsa::(anonymous namespace)::", "").replace("void ", "").split("(")[0]      # noqa: E731
    return {short(d): out[n] for n, d in zip(names, dem)}


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        pytest.fail("visualslam_amd/lib/libvisualslam_amd.so is not built (python __graft_entry__.py build)")
    return _kernels()


def _regs(k):
    return int(k[".vgpr_count"])      # gfx950: the unified count (architectural + accumulation registers)


def test_no_kernel_uses_scratch_memory(kernels):
    assert len(kernels) > 80
    bad = {n: k[".private_segment_fixed_size"] for n, k in kernels.items() if int(k[".private_segment_fixed_size"]) != 0}
    assert not bad, f"kernels with private (scratch) memory: {bad}"


def test_pyrlk_21x21_kernels_take_exactly_104_registers(kernels):
    """four waves per SIMD and no fifth (pyrlk.hip: at 96 or fewer a fifth PyrLK wave crowds out the other streams' short kernels)"""
    found = [n for n in kernels if n.startswith(("k_pyrlk_104<7, 21", "k_pyrlk_cells_104<7, 21"))]
    assert len(found) >= 12
    for n in found:
        assert _regs(kernels[n]) == 104, (n, _regs(kernels[n]))


def test_short_kernels_fit_beside_four_pyrlk_waves(kernels):
    """per SIMD: 512 - 4 x 104 = 96 registers; the merge workgroup puts two waves on each SIMD, the others one"""
    merge = [n for n in kernels if n.startswith("k_merge_compact")]
    assert merge
    for n in merge:
        assert _regs(kernels[n]) <= 48, (n, _regs(kernels[n]))
    for n, k in kernels.items():
        if n.startswith(("k_pyramid_fused", "k_harris_cells", "k_fast_cells", "k_rgbx_to_u8", "k_warp_perspective")):
            assert _regs(k) <= 96, (n, _regs(k))


def test_find_homography_fits_beside_four_pyrlk_waves(kernels):
    """the kernel the graphs run (arrays of up to 4096 correspondences) and the one for arrays of any size, alone and with a
    detector's merge pass as workgroup 0: one wave per SIMD in the 96 registers four PyrLK waves leave, the on-chip kernel's 32 KB
    of points in dynamic LDS (homography.hip: declared statically the compiler rounds the register count up to 97 = 104
    allocated, and a 298-register form cost the 32-stream configs[4] bench a quarter of its frame rate,
    profiles/r03_homography_dummy_kernels.txt)"""
    for n in ("k_find_homography_lean", "k_find_homography_lean_merge", "k_find_homography", "k_find_homography_merge"):
        k = kernels[n]
        assert _regs(k) <= 96, (n, _regs(k))
        assert int(k[".group_segment_fixed_size"]) <= 8 * 1024, (n, k[".group_segment_fixed_size"])
