import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.quick_bench import run
for tune in range(8):
    os.environ["LBM_TUNE"] = str(tune)
    run(256, 256, 256, 1, skew=3072)
os.environ["LBM_TUNE"] = "7"
for skew in (1024, 2048, 3072, 5120, 266496):
    run(256, 256, 256, 1, skew=skew)
run(256, 256, 256, 0, skew=3072)
run(512, 256, 256, 0, skew=3072)
