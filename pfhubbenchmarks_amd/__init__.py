"""pfhubbenchmarks_amd -- MI355X-native hot path of vpuri3/PFHubBenchmarks (PFHub BM1 / BM6 time stepping).

Layout: csrc/ (HIP kernels + C ABI -> libpfhip.so), lib.py (ctypes binding), solver.py (host-side solver objects),
pfbase.py (mirror of the reference's dolfin/pfbase.py helpers used on the hot path), drivers (bench1 / bench6).
"""
__version__ = "0.1.0"
