"""Run pytest against an alternative build of the library: python tools/pytest_with_lib.py <lib.so> <pytest args...>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spath_amd import capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:]))
