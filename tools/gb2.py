import sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import gemm_bench as g
g.SHAPES = [("k4096", 16384, 2048, 4096, 0, 0, 0), ("k2048", 16384, 2048, 2048, 0, 0, 0), ("k512", 16384, 2048, 512, 0, 0, 0), ("k512big", 65536, 2048, 512, 0, 0, 0)]
g.main()
