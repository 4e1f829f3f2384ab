"""Static audit of the compiled gfx950 code (no GPU needed: hipcc cross-compiles to assembly).

The fused MLP kernels issue their LDS fragment reads by hand (inline asm) so that hipcc cannot sink them next
to their use; the price is that the compiler's s_waitcnt bookkeeping does not protect those registers.  The
audit (scripts/audit_asm_loads.py) walks the emitted ISA and fails if any instruction touches the destination
of a hand-issued read before a covering `s_waitcnt lgkmcnt`, or if compiler-generated code uses M0, which the
LDS-DMA pieces overwrite without restoring."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "torch-nerf_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-Wno-unused-function", "-S", "--cuda-device-only"]


@pytest.mark.parametrize("name", ["mlp_forward", "mlp_backward", "mlp_forward_bf16", "render_fused", "mlp_layered"])
def test_hand_issued_reads_are_waited_for(name, tmp_path):
    asm = tmp_path / (name + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, os.path.join(CSRC, name + ".hip"), "-o", str(asm)],
                          stderr=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "audit_asm_loads.py"), str(asm)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "hand-issued LDS reads, 0 problems" in out.stdout
    assert int(out.stdout.rsplit(":", 1)[1].split()[0]) > 100      # the audit really saw the reads
    # no kernel of these files may spill beyond what scratch_allow.txt lists for ITS symbol: a scratch reload is a VMEM load
    # whose vmcnt wait also waits for the weight DMA (scripts/audit_scratch.py; a file-wide allowance would hide a new
    # spill in any of the layered family's ~50 instances)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "audit_scratch.py"), str(asm)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
