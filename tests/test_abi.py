"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/p25fe.h declares,
and its host-only entry points behave (no compute calls without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from p25rx_amd import _lib
    return _lib


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "p25fe.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(p25fe_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (p25fe_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported


def test_rccl_library_exports_its_header(lib):
    """include/p25fe_rccl.h (the N > 1 step behind the C ABI): every declared symbol is exported by libp25fe_rccl.so,
    which links librccl and libp25fe.so -- checked without loading it (no GPU here)."""
    hdr = open(os.path.join(ROOT, "include", "p25fe_rccl.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(p25fe_[a-z0-9_]+)\s*\(", hdr))
    from p25rx_amd import rccl
    assert declared == set(rccl.SYMBOLS)
    so = os.path.join(ROOT, "p25rx_amd", "libp25fe_rccl.so")
    out = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
    assert declared <= set(re.findall(r" T (p25fe_[a-z0-9_]+)", out))
    needed = subprocess.check_output(["readelf", "-d", so]).decode()
    assert "librccl.so" in needed and "libp25fe.so" in needed
    assert os.path.exists(os.path.join(ROOT, "build", "p25fe_shards"))


def test_rust_binding_file_declares_the_whole_abi(lib):
    """bindings/p25fe.rs (uncompiled here: no Rust toolchain) is the `extern "C"` block a p25rx maintainer drops in: it must name
    every symbol of both headers, carry the ABI version, and its #[repr(C)] Config must list the C struct's fields in order."""
    from p25rx_amd import rccl
    rs = open(os.path.join(ROOT, "bindings", "p25fe.rs")).read()
    declared = set(re.findall(r"pub fn (p25fe_[a-z0-9_]+)\(", rs))
    assert declared == set(lib.SYMBOLS) | set(rccl.SYMBOLS), declared ^ (set(lib.SYMBOLS) | set(rccl.SYMBOLS))
    assert "pub const ABI_VERSION: i32 = %d;" % lib.ABI_VERSION in rs
    body = rs[rs.index("pub struct Config {"):]
    body = body[:body.index("}")]
    fields = re.findall(r"pub (\w+):", body)
    assert fields == [f[0] for f in lib.Config._fields_], fields
    hdr = open(os.path.join(ROOT, "include", "p25fe.h")).read()
    cstruct = hdr[hdr.index("typedef struct p25fe_config {"):hdr.index("} p25fe_config_t;")]
    cstruct = re.sub(r"/\*.*?\*/", "", cstruct, flags=re.S)
    cfields = re.findall(r"(\w+)(?:\[[^\]]*\])?\s*[;,]", cstruct)
    assert [f for f in cfields if f not in ("p25fe_config",)] == fields, cfields


def test_default_config_matches_spec(lib, spec):
    cfg = lib.default_config()
    assert cfg.abi_version == lib.ABI_VERSION == 6 and cfg.n_channels == 1 and cfg.symbol_clock == 0
    # ABI 5: Decimator::new(5)'s phase and MovingAverage::new(10) as a table (src/demod.rs:50, 52) default to the build's numbers
    assert cfg.decim_phase == spec["decim_phase"] == 4 and cfg.n_avg_taps == spec["boxcar_len"] == 10
    assert np.array_equal(np.array(cfg.avg_taps[:10], dtype=np.float32), np.array(spec["avg_taps"], dtype=np.float32))
    assert all(np.float32(v) == np.float32(spec["boxcar_scale"]) for v in spec["avg_taps"])
    for bad in (dict(decim_phase=5), dict(decim_phase=-1), dict(avg_taps=[]), dict(avg_taps=[0.1] * 65), dict(avg_taps=[float("nan")])):
        with pytest.raises(lib.P25feError) as e:
            lib.probe_variant(lib.make_config(**bad))
        assert e.value.status == lib.ERR_ARG
    assert lib.probe_variant(lib.make_config(decim_phase=0)) == lib.VARIANT_BUILTIN      # the phase is the host's: the built-in kernels
    # ABI 4: the run-time arguments of the reference's constructors (src/demod.rs:54, 83) default to the build's numbers
    assert cfg.fm_deviation_hz == spec["fm_deviation_hz"] == 5000 and cfg.fm_sample_rate_hz == spec["fm_sample_rate_hz"] == 48000
    assert cfg.fm_gain == 0.0 and cfg.specialize == lib.SPECIALIZE_AUTO and cfg.u8_lut_valid == 0
    assert np.float32(cfg.u8_scale) == np.float32(spec["u8_scale"]) and np.float32(cfg.u8_offset) == np.float32(spec["u8_offset"])
    from oracle import oracle as O
    assert np.float32(O.fm_gain_from(cfg.fm_deviation_hz, cfg.fm_sample_rate_hz)) == np.float32(spec["fm_gain"])
    assert cfg.n_decim_taps == spec["t1"] and cfg.n_chan_taps == spec["t2"]
    assert np.array_equal(np.array(cfg.decim_taps[:spec["t1"]], dtype=np.float32), np.array(spec["decim_taps"], dtype=np.float32))
    assert np.array_equal(np.array(cfg.chan_taps[:spec["t2"]], dtype=np.float32), np.array(spec["chan_taps"], dtype=np.float32))


def _hsaco_kernels(path):
    """{kernel name: (vgprs, scratch bytes per lane)} from a code object's metadata"""
    out = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path]).decode()
    names = re.findall(r"\.name:\s+(\S+)", out)
    scr = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", out)]
    vg = [int(x) for x in re.findall(r"\.vgpr_count:\s+(\d+)", out)]
    assert len(names) == len(scr) == len(vg)
    return {n: (v, s_) for n, v, s_ in zip(names, vg, scr)}


def test_specialize_compiles_without_a_gpu_and_caches(lib, tmp_path):
    """p25fe_specialize (the `make SPEC=` step of p25fe_create's hipRTC path): a caller's tables / constants become the
    immediates of the same kernels; the code object lands under a hash of the numbers; the six entry points exist, none
    of them spills; the build's own numbers need nothing; bad numbers are argument errors."""
    rng = np.random.default_rng(5)
    d = str(tmp_path / "spec")
    assert lib.specialize(lib.default_config(), d) == ""              # the library's own kernels carry these
    dt, ct = (rng.standard_normal(31) * 0.1).astype(np.float32), (rng.standard_normal(41) * 0.1).astype(np.float32)
    cfg = lib.make_config(decim_taps=list(dt), chan_taps=list(ct))
    # three sets of numbers compiled CONCURRENTLY (one process per GPU, or one thread per tuner, may create handles at the same
    # time: hipRTC and the cache's write-then-rename must cope): other 31 / 41 tables; 64 / 64 tables + a non-affine u8 table +
    # another deviation (the long geometry with the LDS table); an affine u8 table given as a TABLE
    import threading
    sc, of = np.float32(1.0 / 128.0), np.float32(-127.0 / 128.0)
    lut = (np.arange(256, dtype=np.float32) * sc + of).astype(np.float32)       # exact in fp32: small integers times 2^-7
    cfgs = [cfg,
            lib.make_config(decim_taps=list((rng.standard_normal(64) * 0.1).astype(np.float32)),
                            chan_taps=list((rng.standard_normal(64) * 0.1).astype(np.float32)),
                            u8_lut=np.tanh((np.arange(256) - 127.5) / 90.0), fm_deviation_hz=4000),
            lib.make_config(u8_lut=lut),
            # ABI 5: a 64-tap post-discriminator filter behind 64 / 64 tables (the fm window in LDS, a 160-output segment halo)
            lib.make_config(decim_taps=list((rng.standard_normal(64) * 0.1).astype(np.float32)),
                            chan_taps=list((rng.standard_normal(64) * 0.1).astype(np.float32)),
                            avg_taps=list((rng.standard_normal(64) * 0.1).astype(np.float32)), decim_phase=2)]
    out = [None] * 4

    def work(i):
        try:
            out[i] = lib.specialize(cfgs[i], d)
        except Exception as e:                                      # noqa: BLE001 -- reported by the assert below
            out[i] = e
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert all(isinstance(o, str) and o for o in out), out
    f1, f64, fa, f5 = out
    assert len({f1, f64, fa, f5}) == 4
    assert os.path.dirname(f1) == d and re.fullmatch(r"p25fe-[0-9a-f]{16}\.hsaco", os.path.basename(f1))
    t1 = os.path.getmtime(f1)
    assert lib.specialize(cfg, d) == f1 and os.path.getmtime(f1) == t1  # second call: found, not rebuilt
    want = {"p25jit_k1_cf32_lin", "p25jit_k1_u8_lin", "p25jit_k1_cf32_pl", "p25jit_k1_u8_pl", "p25jit_chunk_cf32", "p25jit_chunk_u8"}
    for f in (f1, f64, fa, f5):
        k = _hsaco_kernels(f)
        assert set(k) == want, set(k) ^ want
        assert all(s_ == 0 for _, s_ in k.values()), k                 # ScratchSize 0 everywhere
    # a table that IS fma(b, s, o) runs as arithmetic: the same code object as naming (s, o) directly
    assert lib.specialize(lib.make_config(u8_scale=sc, u8_offset=of), d) == fa
    # argument errors: a NaN tap, a zero deviation, an unknown specialize mode, the old ABI
    for bad in (dict(decim_taps=[float("nan")] * 31), dict(fm_deviation_hz=0), dict(specialize=7), dict(fm_gain=float("inf"))):
        with pytest.raises(lib.P25feError) as e:
            lib.specialize(lib.make_config(**bad), d)
        assert e.value.status == lib.ERR_ARG
    old = lib.default_config()
    old.abi_version = 4
    with pytest.raises(lib.P25feError):
        lib.specialize(old, d)


def test_library_loads_and_degrades_without_hiprtc(lib, tmp_path):
    """hipRTC is loaded on first use, not linked: libp25fe.so has no NEEDED entry for it, and a process that cannot find it
    still finds code objects an ahead-of-time run left (the file name does not depend on the compiler) -- only compiling fails,
    loudly (P25FE_ERR_JIT with a log that says why)."""
    import sys
    needed = subprocess.check_output(["readelf", "-d", lib.LIB_PATH]).decode()
    assert "hiprtc" not in needed
    d = str(tmp_path / "aot")
    cfg_src = "from p25rx_amd import _lib; cfg = _lib.make_config(fm_deviation_hz=4321); "
    made = subprocess.check_output([sys.executable, "-c", cfg_src + "print(_lib.specialize(cfg, %r))" % d], cwd=ROOT, text=True).strip()
    assert os.path.exists(made)
    env = dict(os.environ, P25FE_HIPRTC="/nonexistent/libhiprtc.so")
    found = subprocess.check_output([sys.executable, "-c", cfg_src + "print(_lib.specialize(cfg, %r))" % d], cwd=ROOT, text=True, env=env).strip()
    assert found == made                                              # looked up, no compiler needed
    r = subprocess.run([sys.executable, "-c", cfg_src.replace("4321", "4322") + "print(_lib.specialize(cfg, %r))" % d], cwd=ROOT, text=True, env=env,
                       capture_output=True)
    assert r.returncode != 0 and "libhiprtc is not available" in r.stderr and "status -7" in r.stderr


def _spec_run(code, env_extra, cwd=ROOT):
    import sys
    env = dict(os.environ, **env_extra)
    env.pop("P25FE_JIT", None)
    return subprocess.run([sys.executable, "-c", "from p25rx_amd import _lib; cfg = _lib.make_config(fm_deviation_hz=4444); " + code],
                          cwd=cwd, text=True, env=env, capture_output=True)


def test_cache_refuses_directories_that_are_not_private(lib, tmp_path):
    """Code objects are loaded into the GPU process from the cache: a directory somebody else could have written (foreign owner,
    or writable by group / others -- e.g. a pre-created /tmp/p25fe-cache-<uid>) is neither read from nor written to; the
    object is then compiled and used from memory, and the log says why."""
    good = tmp_path / "good"
    made = _spec_run("print(_lib.specialize(cfg, %r))" % str(good), {})
    assert made.returncode == 0, made.stderr
    name = os.path.basename(made.stdout.strip())
    assert (os.stat(good).st_mode & 0o77) == 0                       # created private
    # the same object in a world-writable directory: ignored as a source ...
    bad = tmp_path / "bad"
    bad.mkdir()
    (bad / name).write_bytes((good / name).read_bytes())
    os.chmod(bad / name, 0o600)
    os.chmod(bad, 0o777)
    cache = tmp_path / "cache"
    r = _spec_run("print(_lib.probe_variant(cfg)); print(_lib.specialize_log())", {"P25FE_SPEC_DIR": str(bad), "P25FE_CACHE_DIR": str(cache)})
    assert r.returncode == 0 and r.stdout.split()[0] == str(lib.VARIANT_SPECIALIZED), r.stderr
    assert "writable by group / others" in r.stdout and len(os.listdir(cache)) == 1          # compiled afresh into the private cache
    # ... and as a store: nothing is written there, the object is used from memory
    before = sorted(os.listdir(bad))
    r = _spec_run("print(_lib.probe_variant(cfg)); print(_lib.specialize_log())", {"P25FE_CACHE_DIR": str(bad)})
    assert r.returncode == 0 and r.stdout.split()[0] == str(lib.VARIANT_SPECIALIZED), r.stderr
    assert "used from memory" in r.stdout and sorted(os.listdir(bad)) == before
    # a world-writable FILE in a private directory, and a symbolic link in place of the directory: ignored too
    os.chmod(good / name, 0o666)
    r = _spec_run("print(_lib.probe_variant(cfg)); print(_lib.specialize_log())", {"P25FE_SPEC_DIR": str(good), "P25FE_CACHE_DIR": str(tmp_path / "c2")})
    assert r.returncode == 0 and "not a private regular file" in r.stdout
    os.chmod(good / name, 0o600)
    link = tmp_path / "link"
    os.symlink(good, link)
    r = _spec_run("print(_lib.probe_variant(cfg)); print(_lib.specialize_log())", {"P25FE_SPEC_DIR": str(link), "P25FE_CACHE_DIR": str(tmp_path / "c3")})
    assert r.returncode == 0 and "symbolic links are not followed" in r.stdout
    if os.geteuid() == 0:                                            # (root can stage the foreign-owner case itself)
        own = tmp_path / "foreign"
        own.mkdir(mode=0o700)
        (own / name).write_bytes((good / name).read_bytes())
        os.chown(own, 12345, 12345)
        r = _spec_run("print(_lib.probe_variant(cfg)); print(_lib.specialize_log())", {"P25FE_SPEC_DIR": str(own), "P25FE_CACHE_DIR": str(tmp_path / "c4")})
        assert r.returncode == 0 and "owned by another user" in r.stdout and len(os.listdir(tmp_path / "c4")) == 1


def test_cache_verifies_content_against_the_numbers(lib, tmp_path):
    """Nothing ties a file's NAME to its content but the trailer the library writes: a complete, well-formed code object with
    one flipped bit, or one built for other numbers and renamed, is 'not cached' -- recompiled, never loaded; a deployment's
    $P25FE_SPEC_DIR is never cleaned up by the library."""
    aot = tmp_path / "aot"
    made = _spec_run("print(_lib.specialize(cfg, %r))" % str(aot), {}).stdout.strip()
    other = _spec_run("cfg = _lib.make_config(fm_deviation_hz=4445); print(_lib.specialize(cfg, %r))" % str(aot), {}).stdout.strip()
    assert os.path.exists(made) and os.path.exists(other) and made != other
    good = open(made, "rb").read()
    flipped = bytearray(good)
    flipped[len(good) // 2] ^= 0x10                                  # still a complete ELF: the section table is intact
    for label, content in (("bit flip", bytes(flipped)), ("other numbers", open(other, "rb").read()), ("no trailer", good[:-32])):
        open(made, "wb").write(content)
        cache = tmp_path / ("cache_" + label.replace(" ", "_"))
        r = _spec_run("print(_lib.probe_variant(cfg)); print(_lib.specialize_log())", {"P25FE_SPEC_DIR": str(aot), "P25FE_CACHE_DIR": str(cache)})
        assert r.returncode == 0 and r.stdout.split()[0] == str(lib.VARIANT_SPECIALIZED), (label, r.stderr)
        assert "damaged or built for other numbers" in r.stdout, label
        assert open(made, "rb").read() == content, label              # the deployment's file is left alone
        stored = os.listdir(cache)
        assert len(stored) == 1 and re.fullmatch(r"p25fe-[0-9a-f]{16}-rtc\d+_\d+\.hsaco", stored[0]), stored   # this toolchain's key
    # the versioned key is looked for first: with both present the ahead-of-time file is not even opened
    open(made, "wb").write(good)
    r = _spec_run("print(_lib.probe_variant(cfg)); print(_lib.specialize_log())", {"P25FE_SPEC_DIR": str(aot), "P25FE_CACHE_DIR": str(cache)})
    assert r.stdout.split()[0] == str(lib.VARIANT_SPECIALIZED) and "ignored" not in r.stdout


def test_auto_fallback_to_the_generic_kernels_is_announced(lib, tmp_path):
    """P25FE_SPECIALIZE_AUTO without hipRTC and without an ahead-of-time object ends on kernels 1.3 - 2.4 x slower: said once on
    stderr (not only to callers that poll p25fe_kernel_variant); REQUIRE fails instead; the build's own numbers say nothing."""
    env = {"P25FE_HIPRTC": "/nonexistent/libhiprtc.so", "P25FE_CACHE_DIR": str(tmp_path / "c")}
    r = _spec_run("print(_lib.probe_variant(cfg)); print(_lib.probe_variant(cfg))", env)
    assert r.returncode == 0 and r.stdout.split() == [str(lib.VARIANT_GENERIC)] * 2, r.stderr
    assert r.stderr.count("running the GENERIC kernels") == 1
    r = _spec_run("print(_lib.probe_variant(cfg))", dict(env, P25FE_QUIET="1"))
    assert r.returncode == 0 and "GENERIC" not in r.stderr
    r = _spec_run("cfg = _lib.make_config(fm_deviation_hz=4444, specialize=_lib.SPECIALIZE_REQUIRE); print(_lib.probe_variant(cfg))", env)
    assert r.returncode != 0 and "status -7" in r.stderr
    r = _spec_run("print(_lib.probe_variant(_lib.default_config()))", env)
    assert r.returncode == 0 and r.stdout.split() == [str(lib.VARIANT_BUILTIN)] and "GENERIC" not in r.stderr


def test_generic_kernels_do_not_spill(lib):
    """No k_frontend / k_chunk instantiation of the library may use scratch memory (a spilled FIR loop is ~2x the time)."""
    k = _hsaco_kernels_of_library(lib.LIB_PATH)
    k1 = {n: v for n, v in k.items() if "k_frontend" in n or "k_chunk" in n}
    assert len(k1) >= 20, sorted(k)
    assert all(s_ == 0 for _, s_ in k1.values()), {n: v for n, v in k1.items() if v[1]}


def _hsaco_kernels_of_library(so):
    """metadata of the gfx950 code object bundled in a HIP shared library"""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + os.path.join(d, "k.hsaco")],
                              stderr=subprocess.DEVNULL)
        return _hsaco_kernels(os.path.join(d, "k.hsaco"))


def test_spec_files_regenerate_bit_for_bit():
    before = {p: open(os.path.join(ROOT, p)).read() for p in ("include/p25fe_spec.h", "tests/golden/spec.json")}
    subprocess.check_call(["python", os.path.join(ROOT, "tools", "gen_spec.py")], stdout=subprocess.DEVNULL)
    for p, txt in before.items():
        assert open(os.path.join(ROOT, p)).read() == txt, p


def test_n_baseband_matches_reference_chunking(lib):
    """16384-sample chunks alternate 3276/3277 outputs (src/demod.rs:87-90)."""
    L = lib.load()
    tot, lens = 0, []
    for _ in range(10):
        lens.append(L.p25fe_n_baseband(tot, 16384))
        tot += 16384
    assert set(lens) == {3276, 3277} and sum(lens) == L.p25fe_n_baseband(0, tot) == tot // 5
    assert L.p25fe_n_baseband(0, 4) == 0 and L.p25fe_n_baseband(0, 5) == 1 and L.p25fe_n_baseband(3, 2) == 1


def test_no_device_fails_loudly(lib):
    """Without a GPU the product refuses to run: no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from p25rx_amd.frontend import FrontEnd
    with pytest.raises(lib.P25feError) as e:
        FrontEnd()
    assert e.value.status == lib.ERR_NO_DEVICE


def test_shard_resolve_host_logic(lib):
    """p25fe_shard_resolve is pure host logic: carry the latest anchor, count pre-sync dibits in closed form."""
    L = lib.load()
    R, A = lib.RESULT_DTYPE, lib.ANCHOR_DTYPE
    summ = np.zeros(4, dtype=R)
    bb0 = np.array([0, 1000, 2000, 3000], dtype=np.uint64)
    bbn = np.array([1000, 1000, 1000, 1000], dtype=np.uint64)
    # shard 0: first event decided at 307 (s = 302), 69 dibits after it; shard 1: no sync; shard 2: re-anchors
    summ[0]["first_event"], summ[0]["n_dibits_after_first"], summ[0]["carry_end"] = 307, 69, 308
    summ[0]["anchor_out"] = (302, 0.2, 0.0, -0.2, 1, 10, 1)
    summ[1]["first_event"] = summ[1]["carry_end"] = -1
    summ[2]["first_event"], summ[2]["n_dibits_after_first"], summ[2]["carry_end"] = 2508, 49, 2509
    summ[2]["anchor_out"] = (2503, 0.3, 0.1, -0.1, 1, 10, 1)
    summ[3]["first_event"] = summ[3]["carry_end"] = -1
    anc = np.zeros(4, dtype=A)
    off = np.zeros(5, dtype=np.uint64)                         # n_shards + 1: the last entry is the capture's total
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert L.p25fe_shard_resolve(p(summ), p(bb0), p(bbn), 4, 0, p(anc), p(off)) == 0
    assert anc["valid"].tolist() == [0, 1, 1, 1] and anc["s"].tolist()[1:] == [302, 302, 2503]
    inst = lambda s, lo, hi: len([n for n in range(lo, hi) if n > s and (n - s) % 10 == 0])
    exp1 = 69
    exp2 = exp1 + inst(302, 1000, 2000)
    exp3 = exp2 + inst(302, 2000, 2509) + 49
    assert off.tolist() == [0, exp1, exp2, exp3, exp3 + inst(2503, 3000, 4000)]
    # a lock drop inside shard 1 (MessageReceiver::resync at sample 1500, no sync word after it): the carry of shard 0
    # governs up to 1499, shards 2's own detection re-acquires
    summ[1]["carry_end"] = 1500
    assert L.p25fe_shard_resolve(p(summ), p(bb0), p(bbn), 4, 0, p(anc), p(off)) == 0
    assert anc["valid"].tolist() == [0, 1, 0, 1]
    e2 = exp1 + inst(302, 1000, 1500)
    assert off.tolist() == [0, exp1, e2, e2 + 49, e2 + 49 + inst(2503, 3000, 4000)]
    # tracking clock: shard 2's one detection takes its period from the carry-in (2503 - 302 = 2201 -> 220 symbols; with the
    # positions' quarter-sample fractions -- +1 for the carry, -2 for the detection -- a period of (4 * 2201 - 3) / (4 * 220)),
    # which replaces the 10 / 1 that pass 1 assumed for its 49 dibits
    summ[1]["carry_end"] = -1
    summ[2]["first_seg_end"] = 3000 - 2
    summ[2]["flags"] = 3                                      # FIRST_TRACKS_CARRY | OUT_PERIOD_FROM_CARRY
    summ[0]["anchor_out"]["valid"] = 1 | (1 << 8)             # fraction +1 in bits 8..10 of `valid`
    summ[2]["anchor_out"]["valid"] = 1 | ((-2 & 7) << 8)      # fraction -2 (the shard's one detection is its anchor_out) ...
    summ[2]["reserved"] = -2 & 7                              # ... and its first own detection
    assert L.p25fe_shard_resolve(p(summ), p(bb0), p(bbn), 4, 1, p(anc), p(off)) == 0
    assert (int(anc["period_d"][3]), int(anc["period_n"][3])) == (4 * 2201 - 3, 4 * 220)
    assert int(anc["valid"][3]) == 1 | ((-2 & 7) << 8)
    instq = lambda s, d, n, lo, hi: len([j for j in range(1, 2000) if lo <= s + (j * d) // n < hi and s + (j * d) // n > s + 5])
    t1 = 69                                                   # shard 0 ends 2 samples early: it owns [-2, 998)
    t2 = t1 + instq(302, 10, 1, 998, 1998)
    t3 = t2 + instq(302, 10, 1, 1998, 2509) + 49 - instq(2503, 10, 1, 2509, 2998) + instq(2503, 4 * 2201 - 3, 4 * 220, 2509, 2998)
    assert off.tolist() == [0, t1, t2, t3, t3 + instq(2503, 4 * 2201 - 3, 4 * 220, 2998, 3998)]


def test_cpp_host_driver_built_and_fails_loudly_without_gpu(lib, tmp_path):
    """The C++ DemodTask/RecvTask mirror links only the C ABI; without a device it aborts (reference: panic = abort)."""
    import torch
    exe = os.path.join(ROOT, "build", "p25fe_replay")
    assert os.path.exists(exe)
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    src = tmp_path / "x.u8"
    src.write_bytes(bytes(64))
    r = subprocess.run([exe, "u8", str(src), str(tmp_path / "o")], capture_output=True)
    assert r.returncode != 0 and b"no usable HIP device" in r.stderr
