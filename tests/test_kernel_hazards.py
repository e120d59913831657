"""A hazard hipcc cannot see: the K-major operand fragments are read with ds_read_b64_tr_b16 from INLINE ASM (the builtin
makes hipcc drain the LDS-DMA pipeline in front of every read, gemm_v3.h), so the compiler's own hazard and liveness
logic does not know that such a read completes long after it is issued. A fragment register is dead to the compiler
as soon as the MFMAs that read it are emitted, and it did hand such registers to the next reads; a lab build that
issued eight reads behind eight MFMAs this way produced wrong products in the last four (r02, LAB_NOTES.md section 3). The kernels now
keep the fragments of the last eight MFMAs as asm INPUTS of every such read. This test holds that property on the
generated code: no ds_read_b64_tr_b16 may write a register that one of the preceding eight MFMAs (same wave, no barrier
in between) reads as its A or B operand. CPU only: hipcc -S cross-compiles without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_transpose_read_lands_in_a_recent_mfma_operand(tmp_path):
    asm = str(tmp_path / "kmajor.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "vbnn_amd", "csrc"), "-S", "--cuda-device-only", os.path.join(ROOT, "tests", "kmajor_kernels.hip"),
           "-o", asm]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    text = open(asm).read()
    assert text.count("ds_read_b64_tr_b16") > 100 and text.count("v_mfma_f32_16x16x32_bf16") > 1000      # the kernels are in there
    chk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lds_war.py"), asm, "8"], capture_output=True, text=True)
    assert chk.returncode == 0, chk.stdout[-3000:]


def test_the_checker_flags_the_pattern_it_guards_against(tmp_path):
    """The checker on the shape of the lab build that computed wrong products: transpose reads landing in the A operand
    of the MFMAs just issued -- and silent on the same code with a barrier between or with other registers."""
    chk = os.path.join(ROOT, "tools", "check_lds_war.py")
    bad = """_Z4kernv:                              ; @_Z4kernv
	v_mfma_f32_16x16x32_bf16 v[0:3], v[62:65], v[70:73], v[0:3]
	v_mfma_f32_16x16x32_bf16 v[4:7], v[62:65], v[74:77], v[4:7]
	ds_read_b64_tr_b16 v[62:63], v100 offset:0x4000
	ds_read_b64_tr_b16 v[64:65], v100 offset:0x4800
"""
    ok_barrier = bad.replace("\tds_read_b64_tr_b16 v[62:63]", "\ts_barrier\n\tds_read_b64_tr_b16 v[62:63]")
    ok_regs = bad.replace("v[62:63], v100", "v[80:81], v100").replace("v[64:65], v100", "v[82:83], v100")
    for name, text, want in (("bad", bad, 1), ("barrier", ok_barrier, 0), ("regs", ok_regs, 0)):
        f = tmp_path / f"{name}.s"
        f.write_text(text)
        res = subprocess.run([sys.executable, chk, str(f), "8"], capture_output=True, text=True)
        assert res.returncode == want, (name, res.stdout)
        if want:
            assert "_Z4kernv" in res.stdout and "2 read(s)" in res.stdout


def _device_disassembly(tmp_path):
    """llvm-objdump -d of the gfx950 code object inside the SHIPPED library (what the GPU box loads)."""
    import shutil
    from vbnn_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build() first: libvbnn_hip.so is missing"
    so = str(tmp_path / "libvbnn_hip.so")
    shutil.copy(_lib.LIB_PATH, so)
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    res = subprocess.run([objdump, "--offloading", so], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    cos = sorted(f for f in os.listdir(str(tmp_path)) if "amdgcn" in f and "gfx950" in f)      # one per translation unit
    assert len(cos) >= 4, os.listdir(str(tmp_path))
    dis = str(tmp_path / "device.dis")
    with open(dis, "w") as f:
        for co in cos:
            res = subprocess.run([objdump, "-d", str(tmp_path / co)], stdout=f, stderr=subprocess.PIPE, text=True, timeout=600)
            assert res.returncode == 0, res.stderr[-2000:]
    return dis


def test_no_lds_read_is_outstanding_at_a_barrier_of_the_shipped_kernels(tmp_path):
    """The pipelined loops refill an LDS stage one phase after its last read, which is legal only when the readers'
    ds_reads have RETURNED before the barrier that releases the refill (gemm_v2.h, v2_wait_barrier). r02's rare wrong
    forward of the two-rank rehearsal was hipcc sinking eight MFMAs -- and the lgkmcnt wait in front of them -- below the
    next s_barrier in the DUAL SCHED-0 instantiations of gemm_nt_v2. This holds the property on every kernel of the
    library as built: at no s_barrier may an LDS read be outstanding."""
    dis = _device_disassembly(tmp_path)
    text = open(dis).read()
    assert text.count("s_barrier") > 300 and "gemm_nt_v2" in text and "gemm_nt_v3" in text      # the kernels are in there
    chk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_barrier_lgkm.py"), dis], capture_output=True, text=True)
    assert chk.returncode == 0, chk.stdout[-4000:]


def test_nothing_touches_a_transpose_reads_destination_before_its_wait_in_the_shipped_kernels(tmp_path):
    """The K-major operands' fragments come from ds_read_b64_tr_b16 issued by one inline-asm statement and handed to the compiler by a
    later one, behind the s_waitcnt that retires them: in between hipcc knows nothing of the load in flight, so no instruction may read
    or write its destination registers (a copy or a spill there moves garbage). tools/audit_tr_reads.py holds that on the generated
    code of every such kernel of the library as built (r04: the alternating K steps carry fragment registers across a barrier)."""
    dis = _device_disassembly(tmp_path)
    chk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_tr_reads.py"), "--dis", dis], capture_output=True, text=True)
    assert chk.returncode == 0, chk.stdout[-4000:]
    assert chk.stdout.count("transpose reads") >= 6 and "gemm_nt_v3ILb1ELb1ELb1E5EpiDw" in chk.stdout, chk.stdout[-2000:]


def test_the_barrier_checker_flags_r02s_code_shape(tmp_path):
    """The checker on the instruction shape hipcc generated in r02 (reads issued, barrier, refill DMA, then the wait) in
    both input formats, through a loop's back edge, and silent once the wait precedes the barrier."""
    chk = os.path.join(ROOT, "tools", "check_barrier_lgkm.py")
    bad = """_Z4kernv:
	s_waitcnt lgkmcnt(0)
.LBB0_1:
	s_waitcnt vmcnt(8)
	s_barrier
	buffer_load_dwordx4 v1, s[4:7], s64 offen lds
	ds_read_b128 v[148:151], v176 offset:4096
	ds_read_b128 v[156:159], v176 offset:6144
	s_waitcnt lgkmcnt(1)
	v_mfma_f32_16x16x32_bf16 v[0:3], v[148:151], v[152:155], v[0:3]
	s_cbranch_scc0 .LBB0_1
	s_endpgm
"""
    good = bad.replace("\ts_waitcnt vmcnt(8)\n", "\ts_waitcnt vmcnt(8) lgkmcnt(0)\n")
    dis_bad = """0000000000001000 <_Z4kernv>:
	s_waitcnt vmcnt(8)                                         // 000000001000: BF8C0F78
	s_barrier                                                  // 000000001004: BF8A0000
	ds_read_b128 v[148:151], v176 offset:4096                  // 000000001008: D9FE1000 940000B0
	s_cbranch_scc0 65532                                       // 000000001010: BF84FFFC <_Z4kernv+0x0>
	s_endpgm                                                   // 000000001014: BF810000
"""
    for name, text, want in (("bad.s", bad, 1), ("good.s", good, 0), ("bad.dis", dis_bad, 1)):
        f = tmp_path / name
        f.write_text(text)
        res = subprocess.run([sys.executable, chk, str(f)], capture_output=True, text=True)
        assert res.returncode == want, (name, res.stdout)


def test_exchange_kernels_fit_beside_the_gemm_workgroups_they_overlap():
    """Co-residency, held on the SHIPPED code objects (r05): the data-parallel exchange's data kernels run
    WHILE the backward's two-pass GEMM launches hold every CU -- one 512-thread workgroup per CU, all 160 KiB of LDS, two waves per
    SIMD. A SIMD's register file is 512 VGPRs per lane, allocated in granules of 8: what two GEMM waves leave must hold an
    exchange wave, or the exchange waits for a CU to drain and the next 256-tile launch finds a CU short (two rounds). So: every
    gemm_nt_v3 instantiation allocates at most 232 registers and spills nothing, every k_p2p data kernel at most 48, no LDS, no
    scratch (tools/kernel_regs.py reads the ELF notes)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_regs
    ks = kernel_regs.kernels()
    gran = lambda v: (int(v) + 7) // 8 * 8
    v3 = [k for k in ks if "gemm_nt_v3" in k["name"]]
    p2p = [k for k in ks if "k_p2p_reduce_scatter" in k["name"] or "k_p2p_all_gather" in k["name"]]
    assert len(v3) >= 10 and len(p2p) >= 8, (len(v3), len(p2p))
    worst = max(gran(k["vgpr"]) + gran(k["agpr"]) for k in v3)
    assert worst <= 232, f"a two-pass GEMM allocates {worst} registers: two waves leave {512 - 2 * worst} per SIMD"
    assert all(int(k["spill"]) == 0 and int(k["scratch"]) == 0 for k in v3), [k["name"][:60] for k in v3 if int(k["spill"]) or int(k["scratch"])]
    for k in p2p:
        assert gran(k["vgpr"]) + gran(k["agpr"]) <= 512 - 2 * worst, (k["name"][:80], k["vgpr"])
        assert int(k["lds"]) == 0 and int(k["scratch"]) == 0 and int(k["spill"]) == 0, k
