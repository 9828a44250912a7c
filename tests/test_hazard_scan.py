"""tools/hazard_scan.py on hand-written assembly snippets: the scan must report each class of finding it exists for (the
wait states hipcc does not pad around inline asm) and stay silent on the padded forms.  No compiler,
no GPU."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("hazard_scan", os.path.join(ROOT, "tools", "hazard_scan.py"))
hz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(hz)


def asm(body):
    return "_Z1kv:\n" + body + "\n\ts_endpgm\n"


def kinds(text):
    return sorted(f.split()[0] for f in hz.scan(text))


def test_transcendental_result_read_by_the_next_asm_instruction():
    bad = asm("\tv_exp_f32_e32 v2, v1\n\t;;#ASMSTART\n\tv_cvt_pk_f16_f32 v1, v2, v1\n\t;;#ASMEND")
    assert kinds(bad) == ["T"]                      # the round-3 f16 featurizer bug
    padded = asm("\tv_exp_f32_e32 v2, v1\n\ts_nop 0\n\t;;#ASMSTART\n\tv_cvt_pk_f16_f32 v1, v2, v1\n\t;;#ASMEND")
    assert kinds(padded) == []
    own = asm("\tv_exp_f32_e32 v2, v1\n\ts_nop 0\n\tv_cvt_pk_f16_f32 v1, v2, v1")   # hipcc's own instruction: not asm
    assert kinds(own) == []


def test_asm_write_into_an_mfma_operand_or_result():
    mfma = "\tv_mfma_f32_16x16x32_f16 v[78:81], v[90:93], v[130:133], v[110:113]\n"
    over_c = asm(mfma + "\t;;#ASMSTART\n\tv_fma_mixlo_f16 v112, -v108, 1.0, v82 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t;;#ASMEND")
    assert "W" in kinds(over_c)                     # overwrites the C operand right behind the MFMA
    over_d = asm(mfma + "\ts_nop 3\n\t;;#ASMSTART\n\tv_max_f32 v79, 0, v5\n\t;;#ASMEND")
    assert "M" in kinds(over_d)                     # overwrites its result 4 wait states later (8 needed)
    far = asm(mfma + "\ts_nop 7\n\ts_nop 0\n\t;;#ASMSTART\n\tv_max_f32 v79, 0, v5\n\t;;#ASMEND")
    assert kinds(far) == []
    src_ab = asm(mfma + "\t;;#ASMSTART\n\tv_max_f32 v90, 0, v5\n\t;;#ASMEND")      # A / B are read at issue
    assert kinds(src_ab) == []
    settled = asm(mfma + "\tv_add_f32_e32 v5, v79, v79\n\t;;#ASMSTART\n\tv_max_f32 v6, 0, v79\n\t;;#ASMEND")
    assert kinds(settled) == []                     # hipcc's own reader (padded by hipcc) came first


def test_lo_half_computed_from_the_wrong_operand():
    cvt = "\tv_cvt_pkrtz_f16_f32 v5, v2, v3\n"
    wrong = asm(cvt + "\t;;#ASMSTART\n\tv_fma_mixlo_f16 v0, -v5, 1.0, v0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t;;#ASMEND")
    assert kinds(wrong) == ["S"]                    # the tied-operand miscompile: element 0 instead of element 2
    right = asm(cvt + "\t;;#ASMSTART\n\tv_fma_mixlo_f16 v2, -v5, 1.0, v2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t;;#ASMEND\n"
                "\t;;#ASMSTART\n\tv_fma_mixhi_f16 v2, -v5, 1.0, v3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t;;#ASMEND")
    assert kinds(right) == []
    copied = asm("\tv_mov_b32_e32 v6, v2\n" + cvt +
                 "\t;;#ASMSTART\n\tv_fma_mixlo_f16 v6, -v5, 1.0, v6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t;;#ASMEND")
    assert kinds(copied) == []


def test_dpp_source_written_too_recently():
    bad = asm("\tv_add_f32_e32 v4, v1, v2\n\t;;#ASMSTART\n\tv_add_f32_dpp v4, v4, v4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t;;#ASMEND")
    assert kinds(bad) == ["D"]
    ok = asm("\tv_add_f32_e32 v4, v1, v2\n\t;;#ASMSTART\n\ts_nop 1\n\tv_add_f32_dpp v4, v4, v4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t;;#ASMEND")
    assert kinds(ok) == []


def test_accumulator_passed_between_two_mfma_forms():
    wide = "\tv_mfma_f32_16x16x32_f16 v[94:97], v[154:157], v[252:255], v[94:97]\n"
    narrow = "\tv_mfma_f32_16x16x16_f16 v[94:97], v[208:209], v[246:247], v[94:97]\n"
    assert kinds(asm(wide + narrow)) == ["X"]       # what hipcc made of the mixed contraction (wrong sums on the GPU)
    assert kinds(asm(narrow + wide)) == ["X"]
    assert kinds(asm(wide + "\ts_nop 3\n" + narrow)) == ["X"]
    assert kinds(asm(wide + "\ts_nop 7\n" + narrow)) == []
    assert kinds(asm(wide + wide)) == []            # one form: the hardware forwards the accumulator
    other = "\tv_mfma_f32_16x16x16_f16 v[90:93], v[208:209], v[246:247], v[90:93]\n"
    assert kinds(asm(wide + other)) == []           # no dependence


def test_wider_window_behind_the_32x32_forms():
    wide = "\tv_mfma_f32_32x32x16_f16 v[0:15], v[90:93], v[130:133], v[0:15]\n"
    small = "\tv_mfma_f32_16x16x32_f16 v[0:3], v[90:93], v[130:133], v[0:3]\n"
    reader = "\ts_nop 7\n\ts_nop 1\n\t;;#ASMSTART\n\tv_max_f32 v20, 0, v2\n\t;;#ASMEND"   # 10 wait states later
    assert "M" in kinds(asm(wide + reader))         # 16 passes are still writing
    assert kinds(asm(small + reader)) == []
    assert kinds(asm(wide + "\ts_nop 7\n\ts_nop 7\n\ts_nop 3\n" + reader.split("\n", 2)[2])) == []
