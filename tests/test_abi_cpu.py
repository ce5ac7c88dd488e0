"""CPU-only checks of the product library: it loads, exports every symbol include/voxtral_hip.h
declares, and its host-side helpers (no compute) agree with the oracle.  No GPU calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    hdr = open(os.path.join(ROOT, "include", "voxtral_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vox_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 50
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in voxtral_hip.h but not exported"
    bound = set(pkg._lib.SIGNATURES)
    assert declared == bound, (declared - bound, bound - declared)
    assert L.vox_abi_version() == 1


def test_no_oracle_in_product(pkg):
    """The product path must not import/link the oracle (parity claims are void otherwise)."""
    pdir = os.path.join(ROOT, "voxtral-mini-realtime-rs_amd")
    for dp, _, fs in os.walk(pdir):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "vox_oracle" not in src and "oracle_lib" not in src and "libvox_oracle" not in src, f
    out = os.popen(f"ldd {pkg.build.LIB_PATH}").read()
    assert "oracle" not in out and "libamdhip64" in out


def test_no_kernel_spills_or_scratch(pkg):
    """Every kernel of the shipped gfx950 code objects: no VGPR spilled to scratch, no private segment (dense2_gemm_kernel<4,*> ran with 187-212 spilled
    VGPRs and ~700 B of scratch per lane, a dead fused-attention GEMV instantiation with 93).  Read from the AMDGPU metadata notes of the library's embedded code
    objects (tools/kernel_resources.py) -- no GPU needed.  SGPR spills into VGPR lanes are not memory traffic and are not counted."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_resources import kernel_resources
    rows = kernel_resources(pkg.build.LIB_PATH)
    assert len(rows) >= 300
    names = " ".join(r["demangled"] for r in rows)
    for k in ("decode_engine_kernel", "decode_engine_b16_kernel", "q4_gemm_big_kernel", "dense2_gemm_kernel", "mel_kernel"):
        assert k in names, k
    bad = [(r["demangled"], r.get(".vgpr_spill_count", 0), r.get(".private_segment_fixed_size", 0)) for r in rows
           if r.get(".vgpr_spill_count", 0) or r.get(".private_segment_fixed_size", 0) or str(r.get(".uses_dynamic_stack", False)).lower() == "true"]
    assert not bad, bad
    eng = [r for r in rows if "decode_engine" in r["demangled"]]
    assert eng and all(r[".vgpr_count"] <= 128 for r in eng)      # 14 waves per CU need <= 128 VGPRs


def test_no_packed_fp32_src1_swap(pkg):
    """MI355X, found in round 6 (two contexts on one GPU gave different encoder outputs run to run): v_pk_mul / add / fma_f32 with op_sel taking SRC1's high dword for the low
    lane returns a wrong low lane while another wave of the CU executes MFMAs -- tools/repro/pk_fp32_corun.cpp reproduces it without any library code (profiles/
    r06_pk_fp32_corun.txt).  hipcc forms the encoding by itself from pair arithmetic, so every shipped code object is scanned for it (no GPU needed); a kernel that shows up
    here gets VOX_NO_PK_F32 (csrc/vox_kernels.h) or a formulation that keeps the lanes straight."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_resources import packed_src1_swaps, PK_SRC1_SWAP
    assert PK_SRC1_SWAP.search("v_pk_mul_f32 v[12:13], v[12:13], v[10:11] op_sel:[0,1] op_sel_hi:[0,0]") and PK_SRC1_SWAP.search("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1]")
    assert not PK_SRC1_SWAP.search("v_pk_mul_f32 v[8:9], v[0:1], v[4:5] op_sel:[1,0] op_sel_hi:[0,0]") and not PK_SRC1_SWAP.search("v_pk_fma_f32 v[2:3], v[6:7], v[2:3], v[20:21] op_sel:[0,0,1] op_sel_hi:[1,1,0]")
    assert not PK_SRC1_SWAP.search("v_pk_fma_f32 v[14:15], v[16:17], v[10:11], v[12:13] op_sel_hi:[0,1,1] neg_lo:[0,0,1]")
    assert PK_SRC1_SWAP.search("v_pk_mov_b32 v[2:3], v[70:71], v[70:71] op_sel:[1,0]") and not PK_SRC1_SWAP.search("v_pk_mov_b32 v[2:3], v[70:71], v[72:73] op_sel:[0,1]")
    hits = packed_src1_swaps(pkg.build.LIB_PATH)
    assert not hits, hits[:8]


def test_compute_fails_loudly_without_gpu(pkg):
    n = C.c_int32(); pkg._lib.check(pkg.lib().vox_device_count(C.byref(n)))
    if n.value > 0:
        pytest.skip("GPU present")
    with pytest.raises(pkg.VoxError, match="no HIP device"):
        pkg.Context(0)


def test_host_helpers_match_oracle(pkg, orc):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(255168) * 0.1).astype(np.float32)
    a = pkg.peak_normalize(x, 0.95); b = x.copy(); orc.lib().orc_peak_normalize(b, b.size, 0.95)
    assert (a == b).all() and abs(np.abs(a).max() - 0.95) < 1e-6
    assert (pkg.peak_normalize(np.zeros(8, np.float32)) == 0).all()
    cfg = pkg.PadConfig.voxtral()
    assert (cfg.n_left_pad_tokens, cfg.extra_right_pad_tokens, cfg.samples_per_token(), cfg.left_pad_samples()) == (76, 17, 1280, 97280)
    p = pkg.pad_audio(x)
    assert p.size == 375040 and (p == orc.pad_audio(x)).all()
    assert pkg.audio.num_audio_tokens(p.size) == 293                       # pad.rs:178-218
    for n in (0, 1, 1279, 1280, 12800, 12801, 256000):
        assert cfg.padded_len(n) == orc.pad_audio(np.zeros(n, np.float32)).size
    cc = pkg.ChunkConfig.voxtral()
    assert not pkg.needs_chunking(240000, cc) and pkg.needs_chunking(240001, cc)
    ch = pkg.chunk_audio(np.arange(500000, dtype=np.float32), cc)          # chunk.rs:185-265
    assert [c.start_sample for c in ch] == [0, 240000, 480000] and ch[-1].is_last and ch[-1].samples.size == 20000
    assert pkg.chunk_audio(np.zeros(500000, np.float32), cc.with_overlap(100))[1].start_sample == 224000
    ch = pkg.chunk_audio(np.zeros(256000, np.float32), cc.with_max_frames(1200))   # CLI default, transcribe.rs:55-57
    assert [(c.start_sample, c.end_sample) for c in ch] == [(0, 192000), (192000, 256000)]
    with pytest.raises(pkg.VoxError):
        pkg.chunk_audio(np.zeros(10, np.float32), pkg.ChunkConfig(100, 160, 16000, 100))
    fb = pkg.MelSpectrogram.mel_filterbank(); fo = np.zeros((128, 201), np.float32); orc.lib().orc_mel_filterbank(fo)
    assert (fb == fo).all()
    w = pkg.MelSpectrogram.hann_window(400); wo = np.zeros(400, np.float32); orc.lib().orc_hann_window(400, wo)
    assert (w == wo).all() and abs(w[1] - 6.1690807e-05) < 1e-8              # mel.rs:396-405
    assert pkg.MelSpectrogram(None).num_frames(375040) == 2344 and 99 <= pkg.MelSpectrogram(None).num_frames(16000) <= 101
    for dim, t in ((4, 1.0), (3072, 6.0), (256, 6.0)):
        assert (pkg.TimeEmbedding(dim).embed(t) == orc.time_embedding(t, dim)).all()


def test_gguf_reader_cpu(pkg, orc, tmp_path):
    S = pkg.synth
    w = np.sin(np.arange(32 * 64, dtype=np.float32) * np.float32(0.001) - 1.0).astype(np.float32)
    q = S.quantize_q4_0(w)
    p = str(tmp_path / "t.gguf")
    S.write_gguf(p, [("test.weight", (64, 32), S.GGML_Q4_0, q), ("b", (8,), S.GGML_F32, np.arange(8, dtype=np.float32))])
    r = pkg.GgufReader.open(p)                                             # gguf/tests.rs:280-325
    assert r.version() == 3 and r.tensor_count() == 2 and sorted(r.tensor_names()) == ["b", "test.weight"]
    info = r.tensor_info("test.weight")
    assert info.shape() == [32, 64] and info.dtype() == pkg.gguf.Q4_0 and info.byte_size() == q.size and info.num_elements() == 2048
    assert (r.tensor_data("test.weight") == q).all()
    assert (r.tensor_data("b").view(np.float32) == np.arange(8)).all()
    assert r.tensor_info("nonexistent") is None
    with pytest.raises(pkg.VoxError, match="not found"):
        r.tensor_data("nonexistent")
    r.close()
    bad = tmp_path / "bad.gguf"; bad.write_bytes(b"NOPE" + b"\0" * 64)
    with pytest.raises(pkg.VoxError, match="magic"):
        pkg.GgufReader.open(str(bad))
    v9 = tmp_path / "v9.gguf"; v9.write_bytes(b"GGUF" + (9).to_bytes(4, "little") + b"\0" * 64)
    with pytest.raises(pkg.VoxError, match="version"):
        pkg.GgufReader.open(str(v9))
    with pytest.raises(pkg.VoxError):
        pkg.GgufReader.open(str(tmp_path / "missing.gguf"))
    # v2 accepted (reader.rs:117-120)
    S.write_gguf(str(tmp_path / "v2.gguf"), [("a", (32, 32), S.GGML_Q4_0, S.quantize_q4_0(np.ones(1024, np.float32)))], version=2)
    assert pkg.GgufReader.open(str(tmp_path / "v2.gguf")).version() == 2
    # from_bytes (reader.rs:98-103) and from_shards (loader.rs:101-107): same view of the same image
    img = open(p, "rb").read()
    rb = pkg.GgufReader.from_bytes(img)
    assert rb.tensor_count() == 2 and (rb.tensor_data("test.weight") == q).all() and rb.tensor_info("b").shape() == [8]
    cut = [0, 5, 100, len(img) // 2, len(img)]
    rs = pkg.GgufReader.from_shards([img[a:b] for a, b in zip(cut[:-1], cut[1:])])
    assert rs.tensor_count() == 2 and (rs.tensor_data("test.weight") == q).all() and (rs.tensor_data("b").view(np.float32) == np.arange(8)).all()
    with pytest.raises(pkg.VoxError, match="magic|small"):
        pkg.GgufReader.from_bytes(b"NOPE" + b"\0" * 64)
    with pytest.raises(pkg.VoxError, match="exceeds|parse|read"):
        pkg.GgufReader.from_bytes(img[: len(img) - 40])                      # truncated tensor data


def test_synthetic_gguf_is_readable_by_both(pkg, orc, tmp_path):
    d = pkg.synth.tiny_dims(); p = str(tmp_path / "tiny.gguf"); pkg.synth.write_synthetic_gguf(p, d, seed=3)
    r = pkg.GgufReader.open(p); names = r.tensor_names()
    assert len(names) == len(pkg.synth.tensor_manifest(d))
    g = orc.lib().orc_gguf_open(p.encode())
    for nm in names[::7]:
        info = r.tensor_info(nm); raw = r.tensor_data(nm)
        ptr = orc.lib().orc_gguf_tensor_data(g, nm.encode())
        assert bytes((C.c_uint8 * info.byte_size()).from_address(ptr)) == raw.tobytes()
    orc.lib().orc_gguf_close(g)
    dense = pkg.synth.gguf_dense_f32(p)
    wq = dense[pkg.synth.ENC + ".transformer.layers.0.attention.wq.weight"]
    assert wq.shape == (128, 128) and abs(float(wq.mean())) < 2e-3 and 0.02 < float(wq.std()) < 0.04


def test_integration_md_sys_block_is_complete_and_current(pkg):
    """INTEGRATION.md section 2 is generated from the header: every declared symbol has its Rust declaration and the block is not stale."""
    import subprocess, sys
    hdr = open(os.path.join(ROOT, "include", "voxtral_hip.h")).read()
    declared = set(re.findall(r"\b(vox_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index("BEGIN GENERATED -sys"):doc.index("END GENERATED -sys")]
    for name in sorted(declared):
        assert f"pub fn {name}(" in block, name
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_sys_block.py"), "--check"]).returncode == 0


def _build_abi_smoke(tmp):
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg_dir = os.path.join(root, "voxtral-mini-realtime-rs_amd")
    exe = os.path.join(str(tmp), "abi_smoke")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "abi_smoke.c"), "-o", exe,
           "-L" + pkg_dir, "-lvoxtral_hip", "-lm", "-Wl,-rpath," + pkg_dir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _build_e2e_piecewise(tmp):
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg_dir = os.path.join(root, "voxtral-mini-realtime-rs_amd")
    exe = os.path.join(str(tmp), "e2e_piecewise")
    r = subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "tools", "e2e_piecewise.c"), "-o", exe,
                        "-L" + pkg_dir, "-lvoxtral_hip", "-lm", "-Wl,-rpath," + pkg_dir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_e2e_piecewise_c_program_builds_and_fails_cleanly_without_a_gpu(pkg, tmp_path):
    """tools/e2e_piecewise.c (the reference's e2e-bench decode loop over the device-resident decoder surface) is valid C11 against the header and links; with no
    usable device it stops at vox_ctx_create with the library's message -- no CPU path behind the ABI."""
    import subprocess
    import numpy as np
    exe = _build_e2e_piecewise(tmp_path)
    wav = tmp_path / "x.f32"; np.zeros(1600, np.float32).tofile(str(wav))
    n = __import__("ctypes").c_int32(-1)
    if pkg.lib().vox_device_count(__import__("ctypes").byref(n)) == 0 and n.value > 0:
        return      # a GPU box: the device run is tests/test_gpu_model.py / test_gpu_fullsize.py
    r = subprocess.run([exe, "/nonexistent.gguf", str(wav)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "vox_ctx_create" in r.stderr


def test_header_is_valid_c11_and_links(pkg, tmp_path):
    """include/voxtral_hip.h through a real C compiler: tests/abi_smoke.c (C11, -Wall -Wextra -Werror) includes the header, calls ~25 entry points with the declared
    types and links against libvoxtral_hip.so -- ctypes only checks names, this catches header / implementation signature drift.  Without a GPU the program runs the
    host-only entry points for real and checks that vox_ctx_create fails with VOX_ERR_HIP (no CPU fallback)."""
    import subprocess
    exe = _build_abi_smoke(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host helpers ok" in r.stdout


def test_host_code_under_address_sanitizer(pkg, tmp_path):
    """vox_api.cpp's host side (GGUF reader, padding / chunk arithmetic, argument checks: ~2 000 lines of pointer arithmetic) rebuilt with -fsanitize=address and driven by
    tests/abi_smoke.c's host section, including every truncation of a GGUF image.  Device objects are linked unchanged (ASAN does not instrument gfx950 code)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg_dir = os.path.join(root, "voxtral-mini-realtime-rs_amd"); bdir = os.path.join(pkg_dir, "build")
    if not (os.path.exists(os.path.join(bdir, "vox_kernels.o")) and os.path.exists(os.path.join(bdir, "vox_engine.o"))):
        pytest.skip("device objects not built")
    hipcc = "/opt/rocm/bin/hipcc"; clang = "/opt/rocm/lib/llvm/bin/clang"
    d = str(tmp_path)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address", "-fno-omit-frame-pointer", "-x", "hip", "-c",
                        os.path.join(pkg_dir, "csrc", "vox_api.cpp"), "-o", os.path.join(d, "vox_api.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-o", os.path.join(d, "libvoxtral_hip.so"),
                        os.path.join(bdir, "vox_kernels.o"), os.path.join(bdir, "vox_engine.o"), os.path.join(d, "vox_api.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([clang, "-std=c11", "-g", "-fsanitize=address", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "abi_smoke.c"), "-o",
                        os.path.join(d, "abi_smoke_asan"), "-L" + d, "-lvoxtral_hip", "-lm", "-Wl,-rpath," + d], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", VOX_SMOKE_HOST_ONLY="1")
    r = subprocess.run([os.path.join(d, "abi_smoke_asan")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr, r.stdout + r.stderr[-3000:]
    assert "gguf reader ok" in r.stdout
