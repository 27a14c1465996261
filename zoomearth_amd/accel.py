"""Minimal stand-in for the slice of `accelerate.Accelerator` the reference's inference entry point uses
(/root/reference/src/eval/infer.py:158,165,171,109,250): `.device`, `.process_index`, `.num_processes`,
`.is_main_process`, `.prepare(model, dataloader)`, `.unwrap_model(model)`.

One process per GPU (torchrun / torch.distributed.run sets RANK, LOCAL_RANK, WORLD_SIZE).  The question stream
shards with NO collective on the data path; `broadcast_weights` is the path's only collective: a
one-time broadcast of the packed weight arena from rank 0 over RCCL/xGMI so the checkpoint is read from disk once.
"""
from __future__ import annotations

import os
from collections import defaultdict


def shard_round_robin(n_items: int, rank: int, world: int):
    """accelerate-style batch sharding: item i goes to rank i % world."""
    return list(range(rank, n_items, world))


def shard_by_tile(image_names, rank: int, world: int):
    """Tile-level longest-processing-time packing: all questions of one tile go to one rank so each
    75-MB tile is decoded/uploaded once; tiles are sorted by question count and greedily assigned to the least
    loaded rank.  Returns the question indices of `rank`, grouped by tile, in dataset order inside a tile.
    Deterministic on every rank (no communication needed)."""
    groups = defaultdict(list)
    for i, name in enumerate(image_names):
        groups[name].append(i)
    order = sorted(groups.items(), key=lambda kv: (-len(kv[1]), kv[0]))
    load = [0] * world
    mine = []
    for name, idx in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += len(idx)
        if r == rank:
            mine.extend(idx)
    return mine


class Accelerator:
    def __init__(self, mixed_precision=None, project_dir=None, log_with=None, **kw):
        self.process_index = int(os.environ.get("RANK", "0"))
        self.local_process_index = int(os.environ.get("LOCAL_RANK", "0"))
        self.num_processes = int(os.environ.get("WORLD_SIZE", "1"))
        self.mixed_precision = mixed_precision
        import torch
        # more local ranks than GPUs is allowed (ranks r and r + n_gpus share a GPU): two ranks of 128 chains interleave
        # their prefill and decode phases on one GPU and answer 52.5 questions/s where one rank of 256 chains answers 47.6
        # (DESIGN.md section 8).  Every rank then loads the checkpoint itself: RCCL takes one rank per GPU.
        self.device = torch.device("cuda", self.local_process_index % max(1, torch.cuda.device_count()))
        self._pg = False

    @property
    def is_main_process(self) -> bool:
        return self.process_index == 0

    def unwrap_model(self, model):
        return model

    def prepare(self, model, dataloader):
        """Model: already on its GPU (the engine owns the weights).  Dataloader: sharded by tile."""
        return model, ShardedLoader(dataloader, self.process_index, self.num_processes)

    # ---- the one collective of the path
    def init_process_group(self, backend="nccl"):
        import torch.distributed as dist
        if self.num_processes > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend, rank=self.process_index, world_size=self.num_processes)
            self._pg = True

    def broadcast_weights(self, arena, src: int = 0, mode: str = "broadcast"):
        """arena: uint8 tensor view of the engine's packed weights (Engine.weights_arena()); identical afterwards on
        every rank.  mode "scatter_allgather" (SURVEY.md section 5 / 8e): xGMI is a point-to-point mesh (7 links per GPU),
        so rank `src` first sends a DIFFERENT 1/N of the arena to every peer -- all of its links busy at once -- and
        the ranks then all-gather the pieces, every link of every GPU carrying 1/N of the bytes, instead of pushing
        the whole 7.5 GB down one ring; "broadcast" (default: the form verified on hardware) is the plain collective.
        Prefer `broadcast_engine(engine)`.  Kept for callers that hold the arena view: an arena that came from
        `Engine.weights_arena()` knows its engine, and the engine's derived copies (fragment / FP8 copies, captured decode
        graphs) are invalidated here after the bytes changed -- `weights_arena()` itself has no side effect since round 3, so
        without this the receiving ranks would keep decoding on stale copies.  A plain tensor is just broadcast."""
        if self.num_processes <= 1:
            return
        import torch.distributed as dist
        self.init_process_group("nccl" if arena.is_cuda else "gloo")
        scatter_allgather_broadcast(arena, src, dist) if mode == "scatter_allgather" else dist.broadcast(arena, src=src)
        engine = getattr(arena, "_ze_keepalive", None)
        if engine is not None and hasattr(engine, "weights_invalidate"):
            if arena.is_cuda:
                import torch
                torch.cuda.synchronize(arena.device)
            engine.weights_invalidate()

    def broadcast_engine(self, engine, src: int = 0, mode=None, backend=None) -> float:
        """The whole packed weight arena of `engine` from rank `src` to every rank (in place) and the bookkeeping that goes
        with a rewritten arena (Engine.weights_invalidate); returns the seconds the collective took."""
        return broadcast_engine_weights(engine, self.process_index, self.num_processes, src, mode, backend)

    def wait_for_everyone(self):
        if self.num_processes > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.barrier()


def broadcast_engine_weights(engine, rank: int, world: int, src: int = 0, mode=None, backend=None, force: bool = False) -> float:
    """replaces: every rank's own `from_pretrained` (/root/reference/src/eval/infer.py:147-151 under
    `accelerate launch`): rank `src` has read the checkpoint, the other ranks receive the packed arena -- the path's one
    collective (RCCL over xGMI; `ZE_DIST_BACKEND=gloo` stages through the host, for boxes where ranks share a GPU).
    mode: "broadcast" (default) or "scatter_allgather" (`ZE_BCAST`).  Returns the seconds of the collective."""
    import time

    import torch
    import torch.distributed as dist
    if world <= 1 and not force:  # (force: a one-rank communicator still runs the collective -- the RCCL path at N = 1)
        return 0.0
    backend = backend or os.environ.get("ZE_DIST_BACKEND", "nccl")
    mode = mode or os.environ.get("ZE_BCAST", "broadcast")
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = dict(device_id=engine.device) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    arena = engine.weights_arena()
    if arena.is_cuda:
        torch.cuda.synchronize(arena.device)
    dist.barrier()
    t0 = time.perf_counter()
    if mode == "scatter_allgather":
        scatter_allgather_broadcast(arena, src, dist)
    else:
        dist.broadcast(arena, src=src)
    if arena.is_cuda:
        torch.cuda.synchronize(arena.device)
    dt = time.perf_counter() - t0
    del arena
    engine.weights_invalidate()
    return dt


def scatter_allgather_broadcast(arena, src: int, dist, align: int = 256) -> None:
    """Broadcast of a flat byte tensor as scatter (point-to-point sends of distinct pieces from `src`) + all-gather.
    Piece size is a multiple of `align` bytes; the tail that does not divide evenly goes by a plain (tiny) broadcast."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = arena.numel()
    piece = (n // world) // align * align
    if piece == 0:
        dist.broadcast(arena, src=src)
        return
    body = arena[: piece * world]
    mine = body[rank * piece:(rank + 1) * piece]
    # 1. scatter: `src` keeps its own piece in place and sends piece r to rank r
    if rank == src:
        ops = [dist.P2POp(dist.isend, body[r * piece:(r + 1) * piece], r) for r in range(world) if r != src]
    else:
        ops = [dist.P2POp(dist.irecv, mine, src)]
    if ops:  # (a one-rank communicator -- the RCCL smoke run of tests/test_gpu_bench_contract.py -- has no peer to send to)
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    # 2. all-gather the pieces in place (the piece of rank r already sits at offset r * piece of its own arena)
    if arena.is_cuda:
        dist.all_gather_into_tensor(body, mine)
    else:  # gloo: list form; the own slot receives a copy of itself
        gathered = [body[r * piece:(r + 1) * piece] for r in range(world)]
        own = mine.clone()
        dist.all_gather(gathered, own)
    # 3. the remainder (< world * align bytes)
    if piece * world < n:
        dist.broadcast(arena[piece * world:], src=src)


def tile_groups(image_names, indices):
    """[(tile name, [dataset indices])] of a rank's question indices, in order (shard_by_tile lists a tile's questions together)."""
    groups = []
    for j in indices:
        name = image_names[j]
        if groups and groups[-1][0] == name:
            groups[-1][1].append(j)
        else:
            groups.append((name, [j]))
    return groups


class TileClaims:
    """Work stealing of WHOLE tiles for the drain tail (VERDICT r3 missing #2).

    replaces: the static per-rank shard of the reference (accelerate splits the dataloader once,
    /root/reference/src/eval/infer.py:165-171): ranks that finish early idle while the slowest rank drains.

    The assignment stays the deterministic tile-level LPT packing (`shard_by_tile`: every rank computes every rank's list, no
    communication); what is added is ONE claim flag per (owner rank, position in its list) in a `torch.distributed.TCPStore`.
    A rank claims each of its own tiles, front to back, right before it starts it; a rank whose own list is exhausted scans
    the other ranks' lists from the BACK (the LPT order puts the small tiles last; owner and thief meet in the middle) and
    takes the first tile whose flag it can set.  `store.add(key, 1) == 1` is the atomic test-and-set.  A tile is therefore
    processed exactly once, by whoever claimed it, with no collective and no traffic while every rank is still busy with its
    own list (one store round trip per tile).  The thief reads and uploads the tile itself (tiles are files on shared
    storage); results stay per-rank JSONL files, merged by question_id afterwards (`merge_results`)."""

    def __init__(self, store, rank: int, world: int, lists):
        self.store, self.rank, self.world, self.lists = store, rank, world, lists
        self.stolen = 0
        self.mine = []       # (owner, pos) of every tile this rank has claimed, in claim order
        self.finished = set()  # ... and those whose last record has left for the rank's file (mark_finished)
        # test hook (tests/test_gpu_infer_e2e.py): ZE_TEST_SLOW_RANK="<rank>:<seconds>" makes that rank a straggler -- it pauses
        # before every claim on its OWN list; kept here, behind the claims-only path, not in the entry point's loop
        slow = os.environ.get("ZE_TEST_SLOW_RANK", "")
        self._slow_s = float(slow.split(":")[1]) if slow and int(slow.split(":")[0]) == rank else 0.0

    @staticmethod
    def connect(rank: int, world: int, lists, port_offset: int = 17, timeout_s: float = 600.0):
        import datetime

        import torch.distributed as dist
        host = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("ZE_STEAL_PORT", str(int(os.environ.get("MASTER_PORT", "29500")) + port_offset)))
        store = dist.TCPStore(host, port, world, is_master=(rank == 0), timeout=datetime.timedelta(seconds=timeout_s),
                              wait_for_workers=False)
        return TileClaims(store, rank, world, lists)

    def mark_finished(self, owner: int, pos: int) -> None:
        self.finished.add((owner, pos))

    def claim(self, owner: int, pos: int) -> bool:
        if self._slow_s > 0 and owner == self.rank:
            import time
            time.sleep(self._slow_s)
        won = self.store.add(f"ze_tile/{owner}/{pos}", 1) == 1
        if won:
            self.mine.append((owner, pos))
        return won

    def steal(self):
        """(owner, pos) of a tile of another rank this rank now owns, or None when nothing is left anywhere."""
        victims = sorted((r for r in range(self.world) if r != self.rank), key=lambda r: -len(self.lists[r]))
        for owner in victims:
            for pos in range(len(self.lists[owner]) - 1, -1, -1):
                if self.claim(owner, pos):
                    self.stolen += 1
                    return owner, pos
        return None

    def finish(self, poll_s: float = 0.05, deadline_s: float = None):
        """Every rank says it will make no more claims; rank 0 (the store's host) stays until all have -- for at most
        `deadline_s` seconds (ZE_STEAL_DEADLINE_S, default 1800): a rank that died never says so, and rank 0 then names the
        missing ranks and fails instead of hanging."""
        import time
        # (the per-rank key FIRST: the counter is what releases rank 0, which then takes the store down with it -- a set() behind
        #  the add() raced with that and could fail a rank that had finished its work)
        self.store.set(f"ze_rank_done/{self.rank}", "1")
        self.store.add("ze_ranks_done", 1)
        if self.rank == 0:
            if deadline_s is None:
                deadline_s = float(os.environ.get("ZE_STEAL_DEADLINE_S", "1800"))
            t_end = time.monotonic() + deadline_s
            while self.store.add("ze_ranks_done", 0) < self.world:
                if time.monotonic() > t_end:
                    missing = [r for r in range(self.world) if not self._said_done(r)]
                    raise RuntimeError(f"TileClaims.finish: ranks {missing} did not report within {deadline_s:.0f} s "
                                       f"(crashed or hung); their claimed tiles may be missing from the results")
                time.sleep(poll_s)

    def _said_done(self, r: int) -> bool:
        try:
            return self.store.check([f"ze_rank_done/{r}"])
        except Exception:
            return False

    def abandon(self, finished=()):
        """Error path: this rank will claim nothing more.  Prints the tiles it had claimed that are not in `finished` (their
        records are in no file: a --resume run answers them) and counts itself done so that rank 0 does not wait for it."""
        import sys
        fin = self.finished | set(finished)
        left = [self.lists[o][p][0] for (o, p) in self.mine if (o, p) not in fin]
        if left:
            print(f"[rank {self.rank}] failed with {len(left)} claimed tile(s) unfinished: {left[:8]}"
                  f"{' ...' if len(left) > 8 else ''} -- rerun with --resume", file=sys.stderr, flush=True)
        try:
            self.store.set(f"ze_rank_done/{self.rank}", "1")
            self.store.add("ze_ranks_done", 1)
        except Exception:
            pass


class ShardedLoader:
    """Iterates the batches of this rank.  `dataset` rows need `image_name`; batches are lists of rows."""

    def __init__(self, dataloader, rank: int, world: int, by_tile: bool = True):
        self.dl, self.rank, self.world, self.by_tile = dataloader, rank, world, by_tile

    def _indices(self):
        if getattr(self, "_idx", None) is None:
            self._idx = self._compute_indices()
        return self._idx

    def _compute_indices(self):
        ds = self.dl.dataset
        n = len(ds)
        if self.world <= 1:
            return list(range(n))
        if self.by_tile:
            try:
                names = ds["image_name"] if not isinstance(ds, list) else [r["image_name"] for r in ds]
                return shard_by_tile(list(names), self.rank, self.world)
            except Exception:
                pass
        return shard_round_robin(n, self.rank, self.world)

    def image_names(self):
        """`image_name` of this rank's rows, in iteration order, from the column alone (no row is materialised)."""
        ds = self.dl.dataset
        idx = self._indices()
        try:
            col = [r["image_name"] for r in ds] if isinstance(ds, list) else ds["image_name"]
            return [col[j] for j in idx]
        except Exception:
            return [ds[j]["image_name"] for j in idx]

    def __len__(self):
        bs = getattr(self.dl, "batch_size", 1) or 1
        return (len(self._indices()) + bs - 1) // bs

    def __iter__(self):
        ds = self.dl.dataset
        bs = getattr(self.dl, "batch_size", 1) or 1
        collate = getattr(self.dl, "collate_fn", None) or (lambda x: x)
        idx = self._indices()
        for i in range(0, len(idx), bs):
            yield collate([ds[j] for j in idx[i: i + bs]])


def merge_results(pattern_prefix: str, world: int, out_path: str) -> int:
    """Concatenate results/{exp}{rank}.jsonl of all ranks sorted by question_id into one file for eval.sh.  A question_id is
    kept ONCE (the first record in rank order): an interrupted --steal run that was resumed can hold a question in the
    thief's file and again in its owner's, and eval.py would count it twice."""
    import json
    rows, seen = [], set()
    for r in range(world):
        p = f"{pattern_prefix}{r}.jsonl"
        if os.path.exists(p):
            with open(p, encoding="utf-8") as f:
                for line in f:
                    if not line.strip():
                        continue
                    d = json.loads(line)
                    key = (str(type(d["question_id"])), d["question_id"])
                    if key in seen:
                        continue
                    seen.add(key)
                    rows.append(d)
    rows.sort(key=lambda d: (str(type(d["question_id"])), d["question_id"]))
    with open(out_path, "w", encoding="utf-8") as f:
        for d in rows:
            f.write(json.dumps(d, ensure_ascii=False) + "\n")
    return len(rows)
