import sys
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
import bench_configs as B
B.run_c4()
B.run_c4_draw()
