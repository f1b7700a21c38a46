"""One rank of the data-parallel training-step test on the MI355X (world_size 2, both ranks on cuda:0, gloo for the two
collectives of the step): ONLY rank 1's batch overflows fp16.  Each rank records its local overflow flag (from its own
backward pass), then runs two full ``train_step``s; the test asserts that both ranks skipped step 1 together and ended
with identical master weights, optimizer step counts and loss scales (the skip decision must be global)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank = dist.get_rank()
    from implementation_phd_lab_vision_amd import train
    from oracle import lifting_oracle as lo
    dev = "cuda:0"
    torch.cuda.set_device(0)
    sd = lo.synthetic_head_state_dict(64, 2, 31)
    m = train.TrainableHead(64, 17, 2, precision="fp16")
    m.load_state_dict(sd); m.to(dev); m.eval()
    optim, scaler = train.AdamW(m, lr=1e-4), train.GradScaler(init_scale=2.0 ** 12)
    g = torch.Generator().manual_seed(310 + rank)
    feats, gt = torch.randn(2, 5, 2048, generator=g).abs().to(dev), torch.randn(2, 5, 17, 3, generator=g).to(dev)
    if rank == 1:
        gt = gt * 3.0e4                       # a huge residual: this rank's scaled fp16 gradients saturate
    m.forward_backward(feats, gt, scaler.get_scale())
    torch.cuda.synchronize()
    local_found = int(m._found.item())
    rec = {"local_found": local_found, "skipped": [], "scales": [], "steps": []}
    for _ in range(2):
        _, _, skipped = m.train_step(feats, gt, optim, scaler)
        rec["skipped"].append(bool(skipped)); rec["scales"].append(scaler.get_scale()); rec["steps"].append(optim.step_count)
    rec["flat_master"] = m.flat_master.cpu()
    torch.save(rec, f"{out}.rank{rank}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
