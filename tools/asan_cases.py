"""Parity cases of tests/parity_cases.py on a sanitizer build of the kernel-emulation library (tools/asan_check.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import mini_stark_amd as ms
import parity_cases as pc
EMU = os.environ.get('MS_EMU_LIB', '/tmp/libministark_emu_asan.so')
cache = {}
def mk(field, fresh=False):
    if fresh: return ms.Context(field, lib_path=EMU)
    if field not in cache: cache[field] = ms.Context(field, lib_path=EMU)
    return cache[field]
for field in (0, 1):
    for log_n in (0, 1, 3, 5, 9, 10, 11, 12, 13, 14):
        pc.case_ntt(mk, field, log_n)
    pc.case_coset_lde(mk, field, 4, 8); pc.case_coset_lde(mk, field, 9, 4); pc.case_coset_lde(mk, field, 14, 8)   # the last one: cooperative tiles behind the virtual pass
    if field == 0: pc.case_ntt(mk, field, 19, batch=1)   # 2^10-row cooperative tile (fibers), exec-masked class's host formulas
    for a in [(16, 1, 2, 2), (16, 1, 4, 4), (3, 1, 2, 2), (4096, 1, 2, 2), (6144, 1, 6, 2), (64, pc.EXT[field], 2, 2), (1 << 13, pc.EXT[field], 2, 2)]:
        pc.case_merkle(mk, field, *a, special=True)
    for lpn, ext in ((6, 1), (16, 1), (2, pc.EXT[field])):
        pc.case_merkle_length_sweep(mk, field, lpn, ext)
    for log_n, blowup in ((4, 2), (3, 8), (6, 8), (10, 8)):
        pc.case_prove(mk, field, log_n, blowup)
    pc.case_prove_base_field_deep_points(mk, field)
    pc.case_errors(mk, field)
    pc.case_prove_wide(mk, field, log_n=6, w=64)
    pc.case_general_closure(mk, field)
    pc.case_merkle_prove(mk, field)
# register-only last pass (msntt::RegPassKernel) on the test plan shape: every register radix, forward / inverse / behind the virtual pass
os.environ["MS_NTT_V2_REGPASS"] = "2"
for log_n in (15, 16, 17, 18, 19):
    pc.case_ntt(lambda f, fresh=False: mk(f, fresh=True), 0, log_n, batch=1)
pc.case_coset_lde(lambda f, fresh=False: mk(f, fresh=True), 0, 15, 8)
del os.environ["MS_NTT_V2_REGPASS"]
# fused last sub-round + store: 2^8-row tiles in both modes (2^16 coefficients behind the virtual pass) and the 2^10-row tiles of a plain 2^20-point transform
pc.case_coset_lde(lambda f, fresh=False: mk(f, fresh=True), 0, 16, 8)
pc.case_ntt(lambda f, fresh=False: mk(f, fresh=True), 0, 20, batch=1)
# r03: BabyBear on the three-sub-round tiles (2^19 points: 2^10 x 2^9 plain; 2^17 coefficients behind the virtual pass), the shared-table first pass
# (needs tiles >= workgroups: MS_NTT_COOP_WGS=8), the virtual linear LDE columns, the arithmetic self test, the proof written into a caller buffer
pc.case_ntt(lambda f, fresh=False: mk(f, fresh=True), 1, 19, batch=1)
pc.case_coset_lde(lambda f, fresh=False: mk(f, fresh=True), 1, 17, 8)
os.environ["MS_NTT_COOP_WGS"] = "8"
pc.case_coset_lde(lambda f, fresh=False: mk(f, fresh=True), 0, 17, 8)
del os.environ["MS_NTT_COOP_WGS"]
os.environ["MS_LDE_VIRTUAL"] = "1"
for field in (0, 1):
    pc.case_prove(lambda f, fresh=False: mk(f, fresh=True), field, 6, 8)
del os.environ["MS_LDE_VIRTUAL"]
for field in (0, 1):
    pc.case_arith_selftest(mk, field, nrand=256)
# r05: MS_FLAG_LATENCY (polled results: flags raised by the trees' last launches, the fused rounds and the evaluations; range flag and length word riding along),
# twice on one context, and the fused rounds with several evaluation-side workgroups
for field in (0, 1):
    lat = ms.Context(field, lib_path=EMU, flags=ms.FLAG_ZERO_DISPLAY_EMPTY | ms.FLAG_LATENCY)
    for log_n, blowup in ((3, 8), (9, 8), (9, 8), (11, 2)):
        pc.case_prove(lambda f, fresh=False: lat, field, log_n, blowup, read_big=False)
    pc.case_device_trace_range_check(lambda f, fresh=False: lat, field, lambda a: (a.ctypes.data, a))
    lat.close()
os.environ["MS_FRI_TAIL_MAX"] = "1048576"
pc.case_prove(lambda f, fresh=False: mk(f, fresh=True), 0, 10, 8)
del os.environ["MS_FRI_TAIL_MAX"]
from mini_stark_amd.host import build_host_library
build_host_library()
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_host_mirror as thm
for field in (0, 1):
    thm.check_into_and_slots(ms.Context(field, lib_path=EMU), 31)
print("asan run complete")
