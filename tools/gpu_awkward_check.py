"""The awkward-instances scene (tests/parity.py) through tools/gpu_world_tree_check.check only: a few seconds on the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import gpu_world_tree_check as g
from tests.parity import awkward_instances
ok = g.check("awkward_instances", awkward_instances(), 256, 160)
g.say("AWKWARD", "GREEN" if ok else "RED")
