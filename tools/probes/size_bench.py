import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.layout_bench import run
for dims in ((256, 256, 256), (512, 256, 256), (1024, 256, 128), (1024, 512, 512), (2048, 128, 128), (256, 1024, 128), (256, 256, 1024), (128, 512, 512), (64, 512, 1024)):
    run(*dims, 1, 0, 2, steps=30, warm=5)
