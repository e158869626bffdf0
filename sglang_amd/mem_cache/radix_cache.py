"""Radix-tree prefix cache over token ids; node values are int64 KV slot ids.

Behavioural mirror of /root/reference/python/sglang/srt/mem_cache/radix_cache.py
(RadixKey :59-243, RadixCache :303-840) and base_prefix_cache.py:42-235, written
from scratch for this runtime:

  * same public API: reset / match_prefix / insert / cache_finished_req /
    cache_unfinished_req / evict / inc_lock_ref / dec_lock_ref /
    evictable_size / protected_size / total_size / all_values_flatten;
  * same observable results (matched slot ids, split points, page rounding,
    evictable/protected accounting, eviction order under LRU and friends);
  * implementation differences: edges are kept in one flat `dict` keyed by the
    first page of the edge label, recency uses a deterministic logical clock
    instead of wall time (so replays and TP ranks make identical decisions --
    the reference needs rank_consensus_checker for that), and eviction keeps
    node ids as the final tie-break.

The tree lives on the host; the slot tensors live on the pool device.
Speculative (bigram / EAGLE) keys are out of scope of this path.
"""
from __future__ import annotations

import heapq
import itertools
from array import array
from dataclasses import dataclass, field
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import torch


# ----------------------------------------------------------------------------- params / results
@dataclass
class MatchPrefixParams:
    key: "RadixKey"
    req: Any = None


@dataclass
class MatchResult:
    device_indices: torch.Tensor
    last_device_node: Any
    last_host_node: Any = None
    best_match_node: Any = None
    host_hit_length: int = 0


@dataclass
class InsertParams:
    key: Optional["RadixKey"] = None
    value: Optional[torch.Tensor] = None
    chunked: bool = False
    priority: int = 0


@dataclass
class InsertResult:
    prefix_len: int
    last_device_node: Any = None


@dataclass
class EvictParams:
    num_tokens: int = 0


@dataclass
class EvictResult:
    num_tokens_evicted: int = 0


@dataclass
class IncLockRefResult:
    delta: int = 0


@dataclass
class DecLockRefResult:
    delta: int = 0


# ----------------------------------------------------------------------------- key
def _as_array(tokens) -> array:
    if isinstance(tokens, array):
        return tokens
    return array("q", tokens)


class RadixKey:
    """Token-id sequence plus an optional namespace (`extra_key`, `cache_salt`)."""

    __slots__ = ("token_ids", "extra_key", "cache_salt")

    def __init__(self, token_ids, extra_key: Optional[str] = None, cache_salt: Optional[str] = None):
        self.token_ids = _as_array(token_ids)
        self.extra_key = extra_key
        self.cache_salt = cache_salt or None

    def __len__(self) -> int:
        return len(self.token_ids)

    def __iter__(self):
        return iter(self.token_ids)

    def __getitem__(self, idx: Union[int, slice]) -> "RadixKey":
        if isinstance(idx, int):
            n = len(self)
            if idx < 0:
                idx += n
            if not 0 <= idx < n:
                raise IndexError(f"RadixKey index out of range: {idx}")
            idx = slice(idx, idx + 1)
        start, stop, step = idx.indices(len(self))
        if step != 1:
            raise ValueError("RadixKey slice step must be 1")
        return RadixKey(self.token_ids[start:stop], self.extra_key, self.cache_salt)

    def __repr__(self) -> str:
        head = list(self.token_ids[:10])
        return f"RadixKey(extra_key={self.extra_key!r}, token_ids={head}{'...' if len(self) > 10 else ''})"

    def page_aligned(self, page_size: int) -> "RadixKey":
        if page_size == 1:
            return self
        return self[: len(self) // page_size * page_size]

    def _same_namespace(self, other: "RadixKey") -> None:
        if self.extra_key != other.extra_key or self.cache_salt != other.cache_salt:
            raise ValueError("RadixKey operations need matching extra_key / cache_salt")

    def match(self, other: "RadixKey", page_size: int = 1) -> int:
        """Length of the common prefix, rounded down to a page multiple."""
        self._same_namespace(other)
        a, b = self.token_ids, other.token_ids
        n = min(len(a), len(b))
        if a[:n] == b[:n]:           # one C-level compare covers the full-hit case
            m = n
        else:                         # bisect on slice equality for the first mismatch
            lo, hi = 0, n             # invariant: a[:lo] == b[:lo], a[:hi] != b[:hi]
            while hi - lo > 1:
                mid = (lo + hi) // 2
                if a[lo:mid] == b[lo:mid]:
                    lo = mid
                else:
                    hi = mid
            m = lo
        return m if page_size == 1 else m // page_size * page_size

    def child_key(self, page_size: int = 1):
        """Hashable label of the first page, namespaced (radix_cache.py:217-229)."""
        t = self.token_ids
        plain = t[0] if page_size == 1 else tuple(t[:page_size])
        if self.cache_salt is not None:
            return ((self.extra_key, self.cache_salt), plain)
        return plain if self.extra_key is None else (self.extra_key, plain)


# ----------------------------------------------------------------------------- node
class TreeNode:
    __slots__ = ("id", "children", "parent", "key", "value", "lock_ref", "last_access_time", "creation_time",
                 "hit_count", "priority")
    _ids = itertools.count()

    def __init__(self, clock: int, priority: int = 0):
        self.id = next(TreeNode._ids)
        self.children: Dict[Any, "TreeNode"] = {}
        self.parent: Optional["TreeNode"] = None
        self.key: Optional[RadixKey] = None
        self.value: Optional[torch.Tensor] = None
        self.lock_ref = 0
        self.last_access_time = clock
        self.creation_time = clock
        self.hit_count = 0
        self.priority = priority

    @property
    def evicted(self) -> bool:
        return self.value is None

    def __lt__(self, other: "TreeNode") -> bool:
        return (self.last_access_time, self.id) < (other.last_access_time, other.id)


# eviction orders (evict_policy.py:16-70): smaller priority tuple is evicted first
_EVICTION = {
    "lru": lambda n: (n.last_access_time,),
    "lfu": lambda n: (n.hit_count, n.last_access_time),
    "fifo": lambda n: (n.creation_time,),
    "mru": lambda n: (-n.last_access_time,),
    "filo": lambda n: (-n.creation_time,),
    "priority": lambda n: (n.priority, n.last_access_time),
    "slru": lambda n: (1 if n.hit_count >= 2 else 0, n.last_access_time),
}


# ----------------------------------------------------------------------------- cache
class RadixCache:
    def __init__(self, req_to_token_pool=None, token_to_kv_pool_allocator=None, page_size: int = 1,
                 disable: bool = False, eviction_policy: str = "lru", disable_finished_insert: bool = False):
        self.req_to_token_pool = req_to_token_pool
        self.token_to_kv_pool_allocator = token_to_kv_pool_allocator
        self.page_size = page_size
        self.disable = disable
        self.disable_finished_insert = disable_finished_insert
        self.fast_unfinished_path = True      # tests switch it off to compare with the reference's two-pass form
        self.eviction_policy = eviction_policy.lower()
        if self.eviction_policy not in _EVICTION:
            raise ValueError(f"unknown eviction policy {eviction_policy!r}")
        self._evict_key = _EVICTION[self.eviction_policy]
        dev = getattr(token_to_kv_pool_allocator, "device", None)
        self.device = torch.device(dev) if isinstance(dev, (str, torch.device)) else torch.device("cpu")
        self.evictable_leaves = set()
        self.reset()

    @classmethod
    def create_simulated(cls, disable: bool = False, mock_allocator=None, page_size: int = 1) -> "RadixCache":
        """A cache without memory pools (tests / simulation), radix_cache.py:335-350."""
        return cls(None, mock_allocator, page_size=page_size, disable=disable)

    # ---- clock -------------------------------------------------------------
    def _tick(self) -> int:
        self._clock += 1
        return self._clock

    # ---- public API ----------------------------------------------------------
    def reset(self) -> None:
        self._clock = 0
        self.root_node = TreeNode(self._tick(), priority=-(1 << 62))
        self.root_node.key = RadixKey(array("q"))
        self.root_node.value = []
        self.root_node.lock_ref = 1
        self.evictable_size_ = 0
        self.protected_size_ = 0
        self.evictable_leaves.clear()
        self._empty = torch.empty((0,), dtype=torch.int64, device=self.device)
        self.hit_tokens = 0
        self.query_tokens = 0

    def _no_match(self) -> MatchResult:
        return MatchResult(self._empty, self.root_node, self.root_node, self.root_node)

    def match_prefix(self, params: Union[MatchPrefixParams, RadixKey]) -> MatchResult:
        """Longest cached prefix of the key (page aligned).  May split one node so the
        match ends on a node boundary; refreshes recency along the path."""
        key = params.key if isinstance(params, MatchPrefixParams) else params
        if self.disable or len(key) == 0:
            return self._no_match()
        key = key.page_aligned(self.page_size)
        if len(key) == 0:
            return self._no_match()
        now = self._tick()
        node = self.root_node
        node.last_access_time = now
        pieces: List[torch.Tensor] = []
        while len(key) > 0:
            child = node.children.get(key.child_key(self.page_size))
            if child is None:
                break
            child.last_access_time = now
            common = child.key.match(key, page_size=self.page_size)
            if common < len(child.key):
                node = self._split(child, common)
                pieces.append(node.value)
                break
            pieces.append(child.value)
            node = child
            key = key[common:]
        # one piece is returned as it is (node values are never written in place): no copy kernel per request
        value = (pieces[0] if len(pieces) == 1 else torch.cat(pieces)) if pieces else self._empty
        self.query_tokens += len(params.key if isinstance(params, MatchPrefixParams) else params)
        self.hit_tokens += int(value.numel())
        return MatchResult(value, node, node, node)

    def insert(self, params: InsertParams) -> InsertResult:
        if self.disable:
            return InsertResult(prefix_len=0)
        key = params.key.page_aligned(self.page_size)
        value = params.value
        if value is not None:
            value = value[: len(key)]
        else:  # test / simulation fallback: the token ids double as values
            value = torch.tensor(list(key.token_ids), dtype=torch.int64)
        priority = params.priority or 0
        now = self._tick()
        node = self.root_node
        node.last_access_time = now
        node.priority = max(node.priority, priority)
        matched = 0
        while len(key) > 0:
            child = node.children.get(key.child_key(self.page_size))
            if child is None:
                break
            child.last_access_time = now
            common = child.key.match(key, page_size=self.page_size)
            matched += common
            key = key[common:]
            value = value[common:]
            node = self._split(child, common) if common < len(child.key) else child
            node.priority = max(node.priority, priority)
            if not params.chunked:
                node.hit_count += 1
        if len(key) > 0:
            leaf = TreeNode(self._tick(), priority=priority)
            leaf.parent = node
            leaf.key = key
            leaf.value = value.clone()
            if not params.chunked:
                leaf.hit_count += 1
            node.children[key.child_key(self.page_size)] = leaf
            self.evictable_size_ += len(key)
            self._refresh_leaf(node)
            self._refresh_leaf(leaf)
            node = leaf
        return InsertResult(prefix_len=matched, last_device_node=node)

    def cache_finished_req(self, req, is_insert: bool = True, *, kv_len_to_handle: int) -> None:
        """Donate a finished request's KV slots to the tree (radix_cache.py:459-514)."""
        if self.disable_finished_insert:
            is_insert = False
        alloc = self.token_to_kv_pool_allocator
        row = self.req_to_token_pool.req_to_token[req.req_pool_idx]
        if self.disable:
            alloc.free_segment(row[req.cache_protected_len:kv_len_to_handle], start_pos=req.cache_protected_len)
            return
        fill = req.get_fill_ids() if hasattr(req, "get_fill_ids") else list(req.origin_input_ids) + list(req.output_ids)
        token_ids = fill[:kv_len_to_handle]
        kv_indices = row[: len(token_ids)]
        key = RadixKey(token_ids, getattr(req, "extra_key", None), getattr(req, "cache_salt", None))
        key = key.page_aligned(self.page_size)
        key_len = len(key)
        values = kv_indices[:key_len].to(dtype=torch.int64, copy=True)
        if is_insert:
            res = self.insert(InsertParams(key=key, value=values, priority=getattr(req, "priority", 0) or 0))
            freed_end = res.prefix_len
        else:
            freed_end = key_len
        # slots duplicated by what the tree already held, then the unaligned tail
        alloc.free_segments([
            (kv_indices[req.cache_protected_len:freed_end], req.cache_protected_len),
            (kv_indices[key_len:], key_len),
        ])
        if req.last_node is not None:
            self.dec_lock_ref(req.last_node)

    def cache_unfinished_req(self, req, chunked: bool = False) -> None:
        """Insert the prompt of a request that keeps running; rewrite its req_to_token
        row with the tree-owned slots and move its lock to the new leaf
        (radix_cache.py:516-584)."""
        if self.disable:
            return
        token_ids = req.get_fill_ids()
        kv_indices = self.req_to_token_pool.req_to_token[req.req_pool_idx, : len(token_ids)]
        key = RadixKey(token_ids, getattr(req, "extra_key", None), getattr(req, "cache_salt", None))
        key = key.page_aligned(self.page_size)
        values = kv_indices[: len(key)].to(dtype=torch.int64, copy=True)
        res = self.insert(InsertParams(key=key, value=values, chunked=chunked,
                                       priority=getattr(req, "priority", 0) or 0))
        if self.fast_unfinished_path and res.prefix_len == req.cache_protected_len and len(key) == len(kv_indices):
            # Common case (no other request inserted these tokens first): the tree took the request's own slots
            # for everything behind its locked prefix, so the row of req_to_token is already what the second
            # match_prefix + rewrite of the reference (radix_cache.py:548-584) would produce.  Same observable
            # state -- recency refresh of the path, hit counters, lock move -- without its device copies.
            now = self._tick()
            node = res.last_device_node
            while node is not None:
                node.last_access_time = now
                node = node.parent
            self.query_tokens += len(key)
            self.hit_tokens += len(key)
            req.cache_protected_len = len(key)
            self.dec_lock_ref(req.last_node)
            self.inc_lock_ref(res.last_device_node)
            req.prefix_indices = values
            req.last_node = res.last_device_node
            return
        if res.prefix_len > req.cache_protected_len:
            self.token_to_kv_pool_allocator.free_segment(kv_indices[req.cache_protected_len:res.prefix_len],
                                                         start_pos=req.cache_protected_len)
        m = self.match_prefix(MatchPrefixParams(key=key))
        new_indices, new_last = m.device_indices, m.last_device_node
        assert len(new_indices) == len(key), f"{len(new_indices)=} {len(key)=}"
        self.req_to_token_pool.write((req.req_pool_idx, slice(req.cache_protected_len, len(new_indices))),
                                     new_indices[req.cache_protected_len:].to(torch.int32))
        req.cache_protected_len = len(new_indices)
        self.dec_lock_ref(req.last_node)
        self.inc_lock_ref(new_last)
        if len(new_indices) < len(kv_indices):
            req.prefix_indices = torch.cat([new_indices, kv_indices[len(new_indices):].to(torch.int64)])
        else:
            req.prefix_indices = new_indices
        req.last_node = new_last

    def evict(self, params: Union[EvictParams, int]) -> EvictResult:
        if self.disable:
            return EvictResult()
        want = params.num_tokens if isinstance(params, EvictParams) else int(params)
        heap = [(self._evict_key(n), n.id, n) for n in self.evictable_leaves]
        heapq.heapify(heap)
        done = 0
        while done < want and heap:
            _, _, victim = heapq.heappop(heap)
            self.token_to_kv_pool_allocator.free_segment(victim.value, start_pos=0)
            done += len(victim.value)
            self._remove_leaf(victim)
            parent = victim.parent
            if not parent.children and parent.lock_ref == 0:
                heapq.heappush(heap, (self._evict_key(parent), parent.id, parent))
        return EvictResult(num_tokens_evicted=done)

    def inc_lock_ref(self, node: TreeNode) -> IncLockRefResult:
        if self.disable:
            return IncLockRefResult(0)
        delta = 0
        while node is not self.root_node:
            if node.lock_ref == 0:
                n = len(node.key)
                self.evictable_size_ -= n
                self.protected_size_ += n
                delta -= n
            node.lock_ref += 1
            self._refresh_leaf(node)
            node = node.parent
        return IncLockRefResult(delta)

    def dec_lock_ref(self, node: TreeNode, params=None) -> DecLockRefResult:
        if self.disable or node is None:
            return DecLockRefResult(0)
        delta = 0
        while node is not self.root_node:
            if node.lock_ref == 1:
                n = len(node.key)
                self.evictable_size_ += n
                self.protected_size_ -= n
                delta += n
            node.lock_ref -= 1
            self._refresh_leaf(node)
            assert node.parent is not None, "this request holds a node from another tree"
            node = node.parent
        return DecLockRefResult(delta)

    def evictable_size(self) -> int:
        return self.evictable_size_

    def protected_size(self) -> int:
        return self.protected_size_

    def total_size(self) -> int:
        total, stack = 0, [self.root_node]
        while stack:
            n = stack.pop()
            total += len(n.value)
            stack.extend(c for c in n.children.values() if not c.evicted)
        return total

    def all_values_flatten(self) -> torch.Tensor:
        vals, stack = [], list(self.root_node.children.values())
        while stack:
            n = stack.pop()
            vals.append(n.value)
            stack.extend(n.children.values())
        return torch.cat(vals) if vals else self._empty

    def pretty_print(self) -> None:
        stack = [(self.root_node, 0)]
        while stack:
            n, ind = stack.pop()
            print(" " * ind, len(n.key), list(n.key.token_ids[:10]), f"r={n.lock_ref}")
            stack.extend((c, ind + 2) for c in n.children.values())
        print(f"#tokens: {self.total_size()}")

    # ---- internals -------------------------------------------------------------
    def _split(self, child: TreeNode, at: int) -> TreeNode:
        """Insert a node holding child.key[:at] between child and its parent."""
        upper = TreeNode(self._tick(), priority=child.priority)
        upper.hit_count = child.hit_count
        upper.parent = child.parent
        upper.lock_ref = child.lock_ref
        upper.key = child.key[:at]
        upper.value = child.value[:at].clone()
        upper.parent.children[child.key.child_key(self.page_size)] = upper
        child.key = child.key[at:]
        child.value = child.value[at:].clone()
        child.parent = upper
        upper.children[child.key.child_key(self.page_size)] = child
        return upper

    def _remove_leaf(self, node: TreeNode) -> None:
        removed = node.parent.children.pop(node.key.child_key(self.page_size), None)
        assert removed is node, "parent does not hold this leaf"
        self.evictable_size_ -= len(node.key)
        self.evictable_leaves.discard(node)
        self._refresh_leaf(node.parent)

    def _refresh_leaf(self, node: TreeNode) -> None:
        """Keep `evictable_leaves` = unlocked nodes with no live children."""
        is_leaf = (not node.evicted and node.lock_ref == 0
                   and all(c.evicted for c in node.children.values()))
        if is_leaf and node is not self.root_node:
            self.evictable_leaves.add(node)
        else:
            self.evictable_leaves.discard(node)
