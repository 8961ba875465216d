#!/usr/bin/env python3
"""Write a REFERENCE-FORMAT checkpoint fixture: the imported reference's model_ad (tiny configuration) with the
regenerable fixture parameters, saved the way kfold_train_adversarial.py:222-227 does (ignite's Checkpoint stores
{'net_model': net_model.state_dict()}; torch.save of that mapping).  Runs only in the authoring container."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from oracle import params as P          # noqa: E402
from oracle import tmf_oracle as O      # noqa: E402
from models.mymodel import model_ad     # noqa: E402

KW = dict(dim=32, depth=2, heads=4, dim_head=8, mlp_dim=128)
spec = O.state_spec("model_ad", **KW)
net = model_ad(dropout=0., **KW)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.init_arrays(spec, seed=7).items()}, strict=True)
# one reference train step on CPU so that the checkpoint holds trained-looking buffers (num_batches_tracked = 1 / 2)
mri, pet, y = (torch.from_numpy(a) for a in P.make_inputs(2, (32, 32, 32), seed=1234))
torch.manual_seed(0)
net.train()
lo, dm, dp = net(mri, pet)
path = os.path.join(HERE, "ref_ckpt_ad_tiny.pt")
torch.save({"net_model": {k: v.half() if v.dtype == torch.float32 else v for k, v in net.state_dict().items()}}, path)
net.load_state_dict({k: v.float() if v.dtype == torch.float16 else v for k, v in torch.load(path)["net_model"].items()})
net.eval()
with torch.no_grad():
    out = net(mri, pet)
np.savez_compressed(os.path.join(HERE, "ref_ckpt_ad_tiny_eval.npz"), logits=out[0].numpy(), d_mri=out[1].numpy(),
                    d_pet=out[2].numpy())
print("wrote", path, os.path.getsize(path) // 1024, "KB")
