"""The built device code must not contain the instruction pattern that made round 2's SGPR-wave-index FFN1 epilogue run-dependent
on MI355X: a > 64-bit VMEM store whose data registers are rewritten by the NEXT instruction (hipcc leaves out the wait state when
the store's soffset is an SGPR; gfx950 needs it - profiles/r03_sgpr_epilogue_hazard.md).  CPU test: disassembles libsome_amd.so."""
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'tools'))
import isa_hazard_scan as scan  # noqa: E402

BAD = '''
0000000000001000 <_Z6kernelv>:
	v_cvt_pk_f16_f32 v161, v166, v167                          // 000000001000: D2A000A1 00034EA6
	buffer_store_dwordx4 v[158:161], v146, s[8:11], s13 offen  // 000000001008: E07C1000 0D029E92
	v_cvt_f32_f16_e32 v158, v138                               // 000000001010: 7F3C178A
	s_endpgm                                                   // 000000001014: BF810000
'''
GOOD = BAD.replace('\tv_cvt_f32_f16_e32 v158, v138', '\ts_nop 1                                                    // 00000000100c: BF800001\n\tv_cvt_f32_f16_e32 v158, v138')
OTHER_REG = BAD.replace('v_cvt_f32_f16_e32 v158, v138', 'v_cvt_f32_f16_e32 v162, v138')
NARROW = BAD.replace('buffer_store_dwordx4 v[158:161]', 'buffer_store_dwordx2 v[158:159]')


def test_scanner_recognises_the_pattern():
    hits = scan.scan_text(BAD)
    assert len(hits) == 1 and hits[0]['sgpr_soffset'] and hits[0]['kernel'] == '_Z6kernelv'
    assert scan.scan_text(GOOD) == []                 # a wait state in between
    assert scan.scan_text(OTHER_REG) == []            # next instruction writes another register
    assert scan.scan_text(NARROW) == []               # 64 bits of store data: no hazard
    glob = BAD.replace('buffer_store_dwordx4 v[158:161], v146, s[8:11], s13 offen', 'global_store_dwordx4 v[2:3], v[158:161], off')
    assert len(scan.scan_text(glob)) == 1             # global stores: the data is the second operand
    glob = BAD.replace('buffer_store_dwordx4 v[158:161], v146, s[8:11], s13 offen', 'buffer_store_dwordx4 v[158:161], v146, s[8:11], 0 offen')
    assert len(scan.scan_text(glob)) == 1 and not scan.scan_text(glob)[0]['sgpr_soffset']


def test_scanner_recognises_overlapping_pack_converts():
    bad = '0000000000001000 <_Z1kv>:\n\tv_cvt_scalef32_2xpk16_fp6_f32 v[2:7], v[18:33], v[2:17], v137   // 0: 00\n'
    good = bad.replace('v[2:7]', 'v[34:39]')
    scale = bad.replace('v[2:7], v[18:33], v[2:17], v137', 'v[130:135], v[98:113], v[114:129], v130')
    assert len(scan.scan_text(bad)) == 1 and scan.scan_text(good) == [] and len(scan.scan_text(scale)) == 1


def test_no_unprotected_wide_store_hazard_in_the_built_library():
    from some_amd import _lib
    if not (scan.LLVM / 'llvm-objdump').exists():
        pytest.skip('llvm-objdump not available')
    assert _lib.LIB_PATH.exists(), 'build the library first (python -m some_amd.build)'
    texts = scan.disassemble(_lib.LIB_PATH)
    assert len(texts) >= 12 and sum(t.count('v_mfma_f32_32x32x16_f16') for t in texts) > 1000      # really the device code
    assert sum(t.count('buffer_store_dwordx4') for t in texts) > 50                               # ... including its wide stores
    hits = [h for t in texts for h in scan.scan_text(t)]
    assert hits == [], hits[:3]
