"""A hazard hipcc cannot see: the K-major operand fragments are read with ds_read_b64_tr_b16 from INLINE ASM (the builtin
makes hipcc drain the LDS-DMA pipeline in front of every read, gemm_v3.h), so the compiler's own hazard and liveness
logic does not know that such a read completes long after it is issued. A fragment register is dead to the compiler
as soon as the MFMAs that read it are emitted, and it did hand such registers to the next reads; a lab build that
issued eight reads behind eight MFMAs this way produced wrong products in the last four (r02). The kernels now
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
