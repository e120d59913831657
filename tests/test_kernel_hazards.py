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
