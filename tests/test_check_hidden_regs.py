"""tools/check_hidden_regs.py is what lets fdnn_ppo.hip keep its accumulators in a0..a159 behind the compiler's back: the
Makefile refuses the kernel's object unless the check passes.  The check itself, on hand-made assembly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_hidden_regs.py")

GOOD = """
	.amdhsa_next_free_vgpr 256
	.amdhsa_accum_offset 96
k:
	v_add_u32_e32 v3, v1, v2
	;;#ASMSTART
	v_mfma_i32_32x32x32_i8 a[0:15], v[2:5], v[6:9], a[0:15]
	v_accvgpr_read_b32 v7, a3
	;;#ASMEND
	ds_read_b128 v[0:3], v9
	s_endpgm
"""


def run(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    r = subprocess.run([sys.executable, TOOL, str(p)], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_accumulation_registers_inside_inline_assembly_only(tmp_path):
    rc, out = run(tmp_path, GOOD)
    assert rc == 0 and "0 compiler instructions on accumulation registers" in out and "96 + 160" in out


def test_a_compiler_spill_into_an_accumulation_register_fails_the_build(tmp_path):
    rc, out = run(tmp_path, GOOD.replace("\tds_read_b128 v[0:3], v9\n", "\tv_accvgpr_write_b32 a1, v91\n\tds_read_b128 v[0:3], v9\n"))
    assert rc != 0 and "1 compiler instructions" in out
    rc, out = run(tmp_path, GOOD.replace("\tds_read_b128 v[0:3], v9\n", "\tglobal_load_dword a0, v[58:59], off\n"))
    assert rc != 0
    rc, out = run(tmp_path, GOOD.replace("\tds_read_b128 v[0:3], v9\n", "\tscratch_store_dwordx4 off, a[32:35], off offset:16\n"))
    assert rc != 0


def test_any_other_register_split_fails_the_build(tmp_path):
    rc, out = run(tmp_path, GOOD.replace(".amdhsa_accum_offset 96", ".amdhsa_accum_offset 104"))
    assert rc != 0 and "NOT 96 + 160" in out
    rc, out = run(tmp_path, GOOD.replace(".amdhsa_next_free_vgpr 256", ".amdhsa_next_free_vgpr 248"))
    assert rc != 0


def test_no_kernel_in_the_file_fails(tmp_path):
    rc, _ = run(tmp_path, "\tv_mov_b32 v0, v1\n")
    assert rc != 0
