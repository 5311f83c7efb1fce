#!/usr/bin/env python3
"""Static DPP-hazard check of built solver libraries: tools/check_dpp_hazard.py <lib.so> [...]  (see mpc_collisionavoidance_amd/dpp_check.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_collisionavoidance_amd import dpp_check
sys.exit(dpp_check.main(sys.argv[1:]))
