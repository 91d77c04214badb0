"""Slot sharding across the GPUs of one node.

Every (cell, slot) resource grid is independent, so whole slots are dealt to ranks and the
data path needs no collective.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only to agree on the
timed region and to add up what was processed.
"""


def shard_slots(total_slots, rank, world):
    """Contiguous, balanced split: returns (first_slot, nof_slots) of `rank`."""
    base, extra = divmod(total_slots, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def cell_affine_rank(cell, slot, world, nof_cells):
    """BASELINE config 4 placement: cell c owns ranks {c * world / nof_cells ...} and alternates slots among them;
    with fewer ranks than cells several cells share a rank."""
    if world >= nof_cells:
        per_cell = world // nof_cells
        return (cell * per_cell + slot % per_cell) % world
    return cell % world


def aggregate(dist, device, local_slots, local_samples, local_seconds):
    """(total slots, total IQ samples, max seconds) over all ranks; identity without a process group."""
    import torch
    if dist is None or not dist.is_initialized():
        return local_slots, local_samples, local_seconds
    sums = torch.tensor([float(local_slots), float(local_samples)], dtype=torch.float64, device=device)
    tmax = torch.tensor([float(local_seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return int(round(sums[0].item())), int(round(sums[1].item())), float(tmax.item())


def gather(dist, device, values):
    """Every rank's list of floats, as a list indexed by rank (one all-gather of a few numbers); [values] without a
    process group."""
    import torch
    if dist is None or not dist.is_initialized():
        return [list(map(float, values))]
    mine = torch.tensor(list(map(float, values)), dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [p.tolist() for p in parts]
