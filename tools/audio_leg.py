"""bench.py's config5_audio leg on its own (BASELINE configs[4]: the substitute clip or --wav PATH tiled to 2^22, 10 levels).
usage (GPU box): python tools/audio_leg.py [wav]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.audio_leg(torch, torch.device("cuda:0"), sys.argv[1] if len(sys.argv) > 1 else None), indent=1))
