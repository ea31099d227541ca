#!/usr/bin/env python3
"""bench.py's two shipped-shape pipeline blocks alone (experiment aid).   python tools/pipe8.py [n_points]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 166_667
sc = bench.Scene(N, 1024, 2048, torch.device("cuda:0"))
print(json.dumps(bench.pipeline_block(sc, num_input=6, num_intermediate=50), indent=1))
print(json.dumps(bench.pipeline_images_block(sc, ipl=8), indent=1))
