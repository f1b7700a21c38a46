"""One rank of the CPU rehearsal of the multi-GPU path (gloo, world_size >= 2).  Launched by
tests/test_distributed_cpu.py with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the env."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch  # noqa: E402

from tests.helpers import GatherBackbone, cli_args  # noqa: E402
from implementation_phd_lab_vision_amd import distributed as D  # noqa: E402
from implementation_phd_lab_vision_amd.preprocess_resnet_features import run_extraction  # noqa: E402
from implementation_phd_lab_vision_amd.synthetic import SyntheticClips  # noqa: E402


def main():
    out, n_clips, seq_len, batch, shard, pool, seed, augment, fp16 = sys.argv[1:10]
    augment, fp16 = augment == "1", fp16 == "1"
    ctx = D.init_from_env(use_gpu=False)
    ds = SyntheticClips(int(n_clips), seq_len=int(seq_len), augment=augment)
    args = cli_args(out, seq_len=int(seq_len), batch_size=int(batch), shard_size=int(shard), shuffle_pool=int(pool),
                    shuffle_seed=int(seed), augment=augment, save_fp16=fp16)
    run_extraction(ds, args, GatherBackbone(), torch.device("cpu"), ctx, log=lambda *_: None)
    D.shutdown(ctx)


if __name__ == "__main__":
    main()
