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
