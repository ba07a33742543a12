"""Boundary evidence for the third named caller, the SFM sample's wrapper class (VisionWorks-SFM-0.90-Samples/samples/sfm/SfM.cpp).

It cannot be BUILT here: besides the front end it calls the closed libvisionworks_sfm (nvxSfm*), which the reference does not
hold.  What can be shown without a GPU is that include/ declares every OpenVX / NVX name the file uses: the compiler checks its
syntax (g++ -fsyntax-only, nothing is produced) against include/ + the reference's nvxio headers, with <NVX/sfm/sfm.h> an EMPTY
file, so that every use of the closed library shows up as an undeclared identifier -- and those must be the ONLY errors.  The
same mechanical rename as for the thin-layer callers (cuda* runtime calls -> hip*, NVX_*_CUDA enums ->
NVX_*_HIP) is applied to a temporary copy.  This is evidence about the drop-in boundary, not parity; the SFM front-end's results
are compared with the oracle in tests/test_gpu_ovx.py::test_sfm_front_end_graph_matches_oracle."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SFM = "/root/reference/VisionWorks/VisionWorks-SFM-0.90-Samples"

RENAME = [(r"cuda_runtime\.h", "hip/hip_runtime_api.h"), (r"\bcuda([A-Z])", r"hip\1"), (r"\b(NVX_[A-Z_]*)_CUDA\b", r"\1_HIP")]
CLOSED_SFM = re.compile(r"^(nvxSfm\w+|nvx_sfm_\w+|NVX_SFM_\w+|state_)$")    # state_ is the member of type nvx_sfm_random_state


def _renamed(src, dst):
    text = open(src, encoding="utf-8", errors="replace").read()
    for pat, rep in RENAME:
        text = re.sub(pat, rep, text)
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as f:
        f.write(text)


@pytest.mark.skipif(not os.path.isdir(SFM), reason="the reference tree is not present on this machine")
def test_sfm_wrapper_compiles_against_our_headers_up_to_the_closed_sfm_library(tmp_path):
    t = str(tmp_path)
    _renamed(f"{SFM}/samples/sfm/SfM.cpp", f"{t}/SfM.cpp")
    _renamed(f"{SFM}/nvxio/include/NVX/Utility.hpp", f"{t}/NVX/Utility.hpp")
    os.makedirs(f"{t}/NVX/sfm", exist_ok=True)
    open(f"{t}/NVX/sfm/sfm.h", "w").close()                      # the closed library's header: empty on purpose
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-fmax-errors=0", "-D__HIP_PLATFORM_AMD__", f"-I{t}", f"-I{ROOT}/include",
           f"-I{SFM}/nvxio/include", f"-I{SFM}/samples/sfm", "-I/opt/rocm/include", "-isystem", f"{SFM}/3rdparty/eigen", f"{t}/SfM.cpp"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    errors = [l for l in res.stderr.splitlines() if " error: " in l]
    assert errors, "the empty <NVX/sfm/sfm.h> must leave the nvxSfm* calls undeclared"
    unexpected = []
    for l in errors:
        m = re.search(r"error: ['‘](\w+)['’] (was not declared in this scope|does not name a type)", l)
        if not (m and CLOSED_SFM.match(m.group(1))):
            unexpected.append(l)
    assert not unexpected, "names include/ does not declare:\n" + "\n".join(unexpected)
    names = {re.search(r"error: ['‘](\w+)['’]", l).group(1) for l in errors}
    # the front-end calls (SfM.cpp:961-962, 980-1003, 1077-1078) are among the names that DID resolve
    assert {"nvxSfmFindFundamentalMatNode", "nvxSfmTwoViewTriangulationNode"} <= names
    assert not names & {"vxuColorConvert", "vxuGaussianPyramid", "nvxuHarrisTrack", "vxOpticalFlowPyrLKNode", "nvxHarrisTrackNode",
                        "vxAccessArrayRange", "NVX_READ_ONLY_HIP", "nvxFindHomographyNode"}
