"""Row-sharding of the score matrix over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The path shards by query rows: rank g owns a contiguous block of rows and all I columns.
  * anchor COLUMNS need no communication (every rank holds its rows' entries);
  * anchor ROWS live on whichever rank owns them: each rank packs the ones in its block and ONE
    all-gather (padded to the largest per-rank count) assembles R [Kq x I] on every rank --
    global anchor indices are sorted and blocks are contiguous, so rank order == reference order;
  * every rank then builds the (tiny) pseudo-inverse and E = U.R redundantly and evaluates its own
    rows; per-query results need no merge beyond an ordered concatenation on rank 0.
There is no collective on the per-query data path.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_rows, rank, world):
	"""Contiguous block [start, end) of rank `rank` (sizes differ by at most one)."""
	base, rem = divmod(n_rows, world)
	start = rank * base + min(rank, rem)
	return start, start + base + (1 if rank < rem else 0)


def split_sorted_indices(global_idxs, n_rows, world):
	"""For sorted global row indices: per-rank (local indices within the rank's block)."""
	g = np.asarray(global_idxs, dtype=np.int64)
	out = []
	for r in range(world):
		s, e = shard_bounds(n_rows, r, world)
		sel = g[(g >= s) & (g < e)]
		out.append(sel - s)
	return out


def _default_pack(local_rows, local_idx):
	from . import ops  # HIP gather; raises on CPU tensors (tests inject their own packer for gloo)
	return ops.gather_rows(local_rows, local_idx)


def allgather_rows(local_block, counts, group=None):
	"""local_block [counts[rank] x I] on every rank -> [sum(counts) x I], ranks in order.  One all-gather."""
	world = dist.get_world_size(group)
	rank = dist.get_rank(group)
	cmax = int(max(counts)) if len(counts) else 0
	I = local_block.shape[1]
	if cmax == 0:
		return local_block.new_empty((0, I))
	send = local_block.new_zeros((cmax, I))
	send[:counts[rank]] = local_block
	# RCCL ("nccl") moves device tensors directly over xGMI.  The gloo backend (CPU tests, and the 2-ranks-on-one-GPU rehearsal)
	# cannot: stage through the host there.
	gloo = dist.get_backend(group) == "gloo"
	staged = local_block.is_cuda and gloo
	as_bytes = gloo and local_block.dtype == torch.bfloat16   # gloo moves bytes: it has no bf16 (nor int16) type
	if staged:
		send = send.cpu()
	if as_bytes:
		send = send.view(torch.uint8)
	recv = send.new_empty((world * cmax, send.shape[1]))
	dist.all_gather_into_tensor(recv, send, group=group)
	if as_bytes:
		recv = recv.view(torch.bfloat16)
	if staged:
		recv = recv.to(local_block.device)
	if all(c == cmax for c in counts):
		return recv
	return torch.cat([recv[r * cmax: r * cmax + counts[r]] for r in range(world)], dim=0)


class ShardedScoreMatrix:
	"""This rank's contiguous row block of a [n_rows x I] score matrix."""

	def __init__(self, local_rows, n_rows, rank=None, world=None, group=None, pack=None):
		self.group = group
		self.rank = dist.get_rank(group) if rank is None else rank
		self.world = dist.get_world_size(group) if world is None else world
		self.n_rows = n_rows
		self.start, self.end = shard_bounds(n_rows, self.rank, self.world)
		assert local_rows.shape[0] == self.end - self.start, "local block does not match shard_bounds"
		self.local = local_rows
		self._pack = pack or _default_pack

	def anchor_rows(self, row_idxs):
		"""A[row_idxs, :] on every rank (row_idxs: sorted global indices).  The single collective of the path."""
		per_rank = split_sorted_indices(row_idxs, self.n_rows, self.world)
		mine = per_rank[self.rank]
		block = self._pack(self.local, mine) if len(mine) else self.local.new_empty((0, self.local.shape[1]))
		return allgather_rows(block, [len(p) for p in per_rank], self.group)

	def local_row_ids(self):
		return np.arange(self.start, self.end)


def allgather_anchor_rows(A_train_local, Kq, rank, world, group=None):
	"""bench.py helper: rank r owns anchor rows [r*Kq/world, (r+1)*Kq/world) of the index matrix."""
	s, e = shard_bounds(Kq, rank, world)
	counts = [shard_bounds(Kq, r, world)[1] - shard_bounds(Kq, r, world)[0] for r in range(world)]
	return allgather_rows(A_train_local[s:e].contiguous(), counts, group)


def gather_rows_to_rank0(local, n_rows, group=None):
	"""Ordered concatenation of per-rank [n_local x c] results on rank 0 (None elsewhere): ONE gather (padded to the largest
	block), nothing is delivered to the ranks that would throw it away."""
	world = dist.get_world_size(group)
	rank = dist.get_rank(group)
	counts = [shard_bounds(n_rows, r, world)[1] - shard_bounds(n_rows, r, world)[0] for r in range(world)]
	cmax, c = max(counts), local.shape[1]
	send = local.new_zeros((cmax, c))
	send[:counts[rank]] = local
	staged = local.is_cuda and dist.get_backend(group) == "gloo"   # (see allgather_rows)
	as_bytes = local.dtype == torch.bfloat16 and dist.get_backend(group) == "gloo"
	if staged:
		send = send.cpu()
	if as_bytes:
		send = send.view(torch.uint8)
		c = send.shape[1]
	recv = [send.new_empty((cmax, c)) for _ in range(world)] if rank == 0 else None
	dist.gather(send, recv, dst=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
	if rank != 0:
		return None
	full = torch.cat([recv[r][:counts[r]] for r in range(world)], dim=0)
	if as_bytes:
		full = full.view(torch.bfloat16)
	return full.to(local.device) if staged else full
