"""Time the bench's 3D leg without CPU baseline / 2D leg, printing the NMS trace. Usage: python tools/time_bench3d.py"""
import os, sys, subprocess
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.exit(subprocess.call([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "2", "--warmup", "1", "--size", "512"]))
