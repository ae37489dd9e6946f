"""Worker of tests/test_bench_cpu.py: ONE rank of bench.py's own control flow on a machine without a GPU.

bench.py is run unmodified (``runpy``); what this file changes is test infrastructure around it: the product's ctypes binding
is pointed at the CPU restatement's build of the engine's C ABI (``oracle/liboccoracle_abi.so`` -- the product itself never looks
for it), whose communicator refuses more than one rank, so that ``init_comm`` takes its file-rendezvous branch on every rank,
exactly as it does on a GPU node whose librccl cannot be used; and ``occ_profile``, which times HIP kernels, answers with
zeros."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from occuspytial_amd import _engine, _lib
    _lib.LIB_PATH = os.path.join(ROOT, 'oracle', 'liboccoracle_abi.so')
    _lib._lib = None
    assert _lib.load().occ_device_count() == 0          # this is not the HIP library
    _engine.Engine.profile = lambda self, reps=200: {k: {'launches': 0, 'total_us': 0.0, 'avg_us': 0.0} for k in _lib.KERNEL_KINDS}
    sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
    runpy.run_path(sys.argv[0], run_name='__main__')


if __name__ == '__main__':
    main()
