"""tools/check_touch_regs.py (the static check behind round 5's touch-register fix) on synthetic assembly: it must flag a touch
destination that is rewritten before the wait that retires it, and must not flag the two shapes hipcc emits around a correct touch
(the other lanes' initialisation in a predicated diamond, an out-of-line block behind an unconditional branch)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_touch_regs", os.path.join(ROOT, "tools", "check_touch_regs.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

HEAD = "_Z1kv:\n"
TOUCH = "\t;;#ASMSTART\n\tglobal_load_dword v5, v[2:3], off\n\t;;#ASMEND\n"
RETIRE = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n"


def run(tmp_path, body):
    f = tmp_path / "k.s"
    f.write_text(HEAD + body + "\ts_endpgm\n")
    return chk.check(str(f))


def test_flags_a_rewritten_touch_destination(tmp_path):
    bad = run(tmp_path, TOUCH + "\tv_mov_b32_e32 v9, v5\n\tv_add_u32_e32 v5, 0x60, v1\n" + RETIRE)
    assert len(bad) == 1 and "v_add_u32_e32 v5" in bad[0][4]


def test_clean_sequences_pass(tmp_path):
    assert run(tmp_path, TOUCH + "\tv_add_f32_e32 v7, v1, v2\n" + RETIRE + "\tv_mov_b32_e32 v5, 0\n") == []
    # the ELSE lanes of a predicated diamond initialise the same register: other lanes
    diamond = ("\ts_and_saveexec_b64 s[4:5], s[0:1]\n" + TOUCH + ".LBB0_1:\n\ts_or_saveexec_b64 s[0:1], s[0:1]\n\ts_xor_b64 exec, exec, s[0:1]\n"
               "\tv_mov_b32_e32 v5, 0\n\ts_or_b64 exec, exec, s[0:1]\n" + RETIRE)
    assert run(tmp_path, diamond) == []
    # code behind an unconditional branch is reached from elsewhere
    outlined = TOUCH + "\ts_branch .LBB0_9\n.LBB0_7:\n\tv_mov_b32_e32 v5, v2\n.LBB0_9:\n" + RETIRE
    assert run(tmp_path, outlined) == []
