"""Runs bench.py's sweep measurement against an alternative build of the library: python tools/bench_lib.py <lib.so> [bench args]"""
import os
import runpy
import sys

from lifcal_amd import _capi as capi

lib = sys.argv[1]
capi.LIB_PATH = lib if os.path.isabs(lib) else os.path.join(os.path.dirname(capi.LIB_PATH), lib)
capi.load_library.__defaults__ = (capi.LIB_PATH,)
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
