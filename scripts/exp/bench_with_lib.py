"""bench.py against a variant build of the library (WR_LIB=path): A/B runs of build flags"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from whisprrec_amd import abi
if os.environ.get("WR_LIB"):
    abi.LIB_PATH = os.path.abspath(os.environ["WR_LIB"])
import bench
bench.main(sys.argv[1:])
