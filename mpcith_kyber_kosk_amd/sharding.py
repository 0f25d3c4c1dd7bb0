"""Multi-GPU sharding of the KOSK path.

Proofs are independent units: rank r of W proves/verifies a contiguous block of the batch with no
data-path exchange.  The only collective is an all-gather (RCCL over xGMI on GPUs, gloo in CPU tests)
of per-rank digest tables -- 32 bytes per party lane (commitment digests, BASELINE.json configs[3])
or per proof (result gather) -- laid out so that the gathered table is indexed by GLOBAL lane/proof id.
"""
from typing import List, Tuple

PARTIES = 1454


def proof_partition(total: int, world: int) -> List[Tuple[int, int]]:
    """(first, count) per rank; contiguous blocks, remainder spread over the low ranks."""
    base, rem = divmod(total, world)
    out, first = [], 0
    for r in range(world):
        cnt = base + (1 if r < rem else 0)
        out.append((first, cnt))
        first += cnt
    return out


def lanes_to_proofs(n_lanes: int) -> int:
    """'N simulated parties' of BASELINE.json -> independent proofs in flight."""
    return -(-n_lanes // PARTIES)


def aligned_partition(n_lanes: int, world: int) -> Tuple[int, int]:
    """Proof-aligned padding for the lane-sharded wording of configs[3]: returns (proofs_per_rank,
    padded_total_proofs) so that no proof straddles a rank boundary (2^20 lanes, 8 ranks -> 91, 728)."""
    proofs = lanes_to_proofs(n_lanes)
    per = -(-proofs // world)
    return per, per * world


def allgather_digest_table(local, world: int, dist=None):
    """local: uint8 tensor [units_per_rank, 32] (same shape on every rank).  Returns [world*units, 32]
    ordered by rank, i.e. by global unit id for the contiguous partitions above."""
    import torch
    if world == 1 or dist is None:
        return local.clone()
    out = torch.empty((world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out
