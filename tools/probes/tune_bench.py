import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.layout_bench import run
scheme = int(sys.argv[1]) if len(sys.argv) > 1 else 0
layout = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for preset in (1, 0):
    run(256, 256, 256, preset, scheme, layout)
