"""bench.py under a given library variant: python tools/dev/benchwith.py <lib.so|-> [bench.py arguments]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uno_amd import _native
if sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
