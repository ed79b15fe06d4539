"""One library, many translation units: an inline function (a class's member functions, a header's helpers) defined with the
same name in two of them is ONE function to the linker, whichever definition it keeps. Round 4 had two different `WorkerPool`
classes in runtime.hip and hs_facade.cpp for an hour: the runtime's pool was destroyed by the facade's destructor (abort in
hsgpu_scratch_free, found by the GPU suite). This test reads the objects the library was linked from and refuses weak
symbols of our own that differ in size between objects."""
import collections
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "hyperscan_amd", "lib", "obj")


def test_no_weak_symbol_of_ours_differs_between_objects():
    objs = sorted(glob.glob(os.path.join(OBJ, "*.o")))
    if len(objs) < 5:
        pytest.skip("the library's objects are not here (they do not travel to the GPU box)")
    seen = collections.defaultdict(dict)
    for o in objs:
        out = subprocess.run(["nm", "-S", "--defined-only", o], capture_output=True, text=True).stdout
        for line in out.splitlines():
            m = re.match(r"^[0-9a-f]+ ([0-9a-f]+) ([WV]) (\S+)$", line)
            if m:
                seen[m.group(3)][os.path.basename(o)] = int(m.group(1), 16)
    ours = {}
    for name, by_obj in seen.items():
        if len(set(by_obj.values())) < 2:
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        # the standard library's and HIP's own inline functions may be compiled differently per unit (inlining decisions
        # inside them); what is ours has no business differing
        if re.match(r"^(std::|__gnu_cxx::|void std::|hip|__hip|operator|typeinfo|vtable|guard variable)", dem) or "std::" in dem.split("(")[0]:
            continue
        ours[dem] = by_obj
    assert not ours, "inline functions of one name and different bodies in different translation units: %r" % ours


def test_filter_kernels_own_no_static_lds():
    """Every hwlm_filter_kernel addresses its dynamic LDS absolutely (the filter image starts at LDS address 0; the kernel traps
    when it does not). Anything that gives such a kernel a STATIC LDS variable moves the dynamic part behind it: round 5's solo
    tail called __syncthreads_or, whose device-library reduction owns an LDS word -- every fused kernel trapped on the GPU
    (HSA_STATUS_ERROR_EXCEPTION), found only there. The objects' metadata says it here: group_segment_fixed_size must be 0."""
    import shutil
    import tempfile

    objs = sorted(glob.glob(os.path.join(OBJ, "scan_*.hip.o")))
    tools = "/opt/rocm/lib/llvm/bin"
    if len(objs) < 5 or not os.path.exists(os.path.join(tools, "llvm-readelf")):
        pytest.skip("the library's objects (or the LLVM tools) are not here")
    seen = 0
    for o in objs:
        d = tempfile.mkdtemp()
        try:
            shutil.copy(o, os.path.join(d, "x.o"))
            subprocess.run([os.path.join(tools, "llvm-objdump"), "--offloading", "x.o"], cwd=d, capture_output=True)
            cos = [f for f in os.listdir(d) if "gfx950" in f]
            if not cos:
                continue
            notes = subprocess.run([os.path.join(tools, "llvm-readelf"), "--notes", os.path.join(d, cos[0])], capture_output=True, text=True).stdout
            lds = None
            for line in notes.splitlines():
                m = re.search(r"\.group_segment_fixed_size:\s*(\d+)", line)
                if m:
                    lds = int(m.group(1))
                m = re.search(r"\.name:\s*(\S+)", line)
                if m and "hwlm_filter_kernel" in m.group(1) and lds is not None:
                    seen += 1
                    assert lds == 0, f"{os.path.basename(o)}: {m.group(1)} has {lds} bytes of static LDS"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    assert seen >= 20
