"""The C-ABI library loads and exports every symbol include/pfhip.h declares; pure-host entry points behave.
No compute call is made here (no GPU in this suite)."""
import ctypes as C
import os
import re

import pytest

from pfhubbenchmarks_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pfhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(pfk?_\w+)\s*\(", text, flags=re.M)))


def test_header_symbols_are_bound_and_exported():
    names = _declared_symbols()
    assert len(names) >= 24
    assert set(names) == set(L.SYMBOLS), (set(names) ^ set(L.SYMBOLS))
    lib = L.load()
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.pf_version() == 100


def test_struct_layouts_match_header():
    cfg = L.default_config(3, 64, 1.0)
    assert cfg.struct_bytes == C.sizeof(L.PfConfig)          # checked again inside pf_create (resolve)
    assert (cfg.rho_s, cfg.c_alpha, cfg.c_beta, cfg.kappa, cfg.M) == (5.0, 0.3, 0.7, 2.0, 5.0)   # bench1.py:32-36
    assert (cfg.k, cfg.eps_r) == (0.09, 90.0)                                                     # bench6.py:38-39
    assert list(cfg.n) == [64, 64, 64] and cfg.nranks == 1


def test_slab_partition():
    lib = L.load()
    f, c = C.c_int(), C.c_int()
    for n, w in [(1024, 8), (512, 2), (10, 3), (7, 7)]:
        tot, nxt = 0, 0
        for r in range(w):
            assert lib.pf_slab_partition(n, w, r, C.byref(f), C.byref(c)) == 0
            assert f.value == nxt
            nxt += c.value
            tot += c.value
            assert c.value in (n // w, n // w + 1)
        assert tot == n
    assert lib.pf_slab_partition(8, 2, 2, C.byref(f), C.byref(c)) == L.PF_ERR_INVALID


def test_field_sizes():
    lib = L.load()
    cfg = L.default_config(3, 32, 1.0)
    assert lib.pf_field_elems(C.byref(cfg)) == 32 ** 3
    assert lib.pf_field_elems_with_ghosts(C.byref(cfg)) == 32 ** 3          # nranks == 1: no ghosts
    cfg.nranks, cfg.rank = 4, 1
    assert lib.pf_field_elems(C.byref(cfg)) == 32 * 32 * 8
    assert lib.pf_field_elems_with_ghosts(C.byref(cfg)) == 32 * 32 * 12
    m = L.default_config(2, 101, 2.0)
    m.bc = L.PF_BC_MIRROR
    assert lib.pf_field_elems(C.byref(m)) == 101 * 101                      # nodes of the physical domain
    assert lib.pf_field_elems_with_ghosts(C.byref(m)) == 200 * 200          # even extension


def test_bad_configs_are_rejected_before_any_hip_call():
    lib = L.load()
    h = C.c_void_p()
    cfg = L.default_config(2, 64, 1.0)
    cfg.struct_bytes = 8
    assert lib.pf_create(C.byref(cfg), C.byref(h)) == L.PF_ERR_INVALID
    assert b"struct_bytes" in lib.pf_last_error(None)
    cfg = L.default_config(2, 64, 1.0)
    cfg.dim = 4
    assert lib.pf_create(C.byref(cfg), C.byref(h)) == L.PF_ERR_INVALID
    cfg = L.default_config(2, 64, 1.0)
    cfg.nranks, cfg.rank = 2, 0
    assert lib.pf_create(C.byref(cfg), C.byref(h)) == L.PF_ERR_UNSUPPORTED   # slabs are 3-D only
    assert not h


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = L.load()
    h = C.c_void_p()
    cfg = L.default_config(2, 64, 1.0)
    rc = lib.pf_create(C.byref(cfg), C.byref(h))
    assert rc == L.PF_ERR_HIP and not h          # no CPU fallback: the handle cannot be created
    assert lib.pf_last_error(None)


def test_integration_md_ctypes_stub_matches_the_abi():
    """the binding a reference maintainer is told to paste must agree with pf_config"""
    import ctypes as C
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n# pfhip_binding.py\n(.*?)```", text, re.S).group(1)
    code = code.replace('C.CDLL("libpfhip.so")',
                        'C.CDLL(%r)' % os.path.join(root, "pfhubbenchmarks_amd", "csrc", "libpfhip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    cfg = ns["PfConfig"]()
    assert ns["lib"].pf_config_default(C.byref(cfg), 2, 201, C.c_double(1.0)) == 0
    assert C.sizeof(cfg) == cfg.struct_bytes
    assert (cfg.dim, cfg.n[0], cfg.n[2], cfg.kappa, cfg.M, cfg.k, cfg.eps_r) == (2, 201, 1, 2.0, 5.0, 0.09, 90.0)


def test_ext_buffer_offsets_follow_the_placement_rule():
    """pf_ext_buffer_offset (host logic only): c[1] sits 64 KB and phi -64 KB (mod 512 KB) from c[0], beyond the buffer"""
    import ctypes as C
    from pfhubbenchmarks_amd import lib as L
    lib = L.load()
    for n, nranks in ((512, 1), (200, 1), (96, 3), (34, 2)):
        cfg = L.default_config(3, n, 1.0)
        cfg.nranks, cfg.rank = nranks, 0
        elems = lib.pf_field_elems_with_ghosts(C.byref(cfg))
        o1, o2 = lib.pf_ext_buffer_offset(C.byref(cfg), 1), lib.pf_ext_buffer_offset(C.byref(cfg), 2)
        assert o1 >= elems and o2 >= o1 + elems
        assert (o1 * 8) % (512 * 1024) == 64 * 1024 and (o2 * 8) % (512 * 1024) == 448 * 1024
        assert o1 * 8 < elems * 8 + 576 * 1024 + 64 * 1024          # no more than one period of padding
    assert lib.pf_ext_buffer_offset(C.byref(cfg), 3) < 0


def test_every_environment_switch_of_the_library_is_documented():
    """The A/B switches read by the library (getenv("PFHIP_...") in csrc/) are all named in DESIGN.md."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in glob.glob(os.path.join(root, "pfhubbenchmarks_amd", "csrc", "*.hip")):
        names |= set(re.findall(r'getenv\("(PFHIP_[A-Z0-9_]+)"\)', open(f).read()))
    design = open(os.path.join(root, "DESIGN.md")).read()
    assert len(names) >= 10
    missing = sorted(n for n in names if n not in design)
    assert not missing, missing
