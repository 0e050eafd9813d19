import sys
sys.path.insert(0, ".")
from cornell_moe_amd import api
api.kxx_build_probe(print)
