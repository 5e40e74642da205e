"""Cache of language-goal embeddings in front of the (frozen) text encoder: ``AdvancedLangEmbeddingBuffer`` (mode/utils/lang_buffer.py:6-71),
the producer of the ``latent_goal`` tensor the denoiser consumes (mode_agent.py:132, 537, 590).

Same public surface and behaviour as the reference — encoder called only on texts never seen, FIFO eviction at capacity (a hit does not refresh
an entry), stacked result in request order, the reference's error fallback, pickle save / load keeping the newest entries — with the storage
laid out for the device: all embeddings live in ONE table ``[capacity, *embedding_shape]`` on the encoder's device and a batch is one
``index_select`` over it instead of a Python list of per-text tensors fed to ``torch.stack`` (a rollout step with B environments is one launch,
and the table is the only allocation).  The encoder itself (CLIP text tower in the reference) is the caller's: any callable
``list[str] -> Tensor[len, ...]``; an ``output_dim`` attribute is only needed for the error fallback, as in the reference."""
from __future__ import annotations

import pickle
import threading
from collections import OrderedDict
from typing import Callable, Iterable, List, Optional, Sequence, Union

import torch


class AdvancedLangEmbeddingBuffer:
    def __init__(self, language_encoder: Callable[[List[str]], torch.Tensor], goal_instruction_buffer_size: int = 10000):
        self.language_encoder = language_encoder
        self.goal_instruction_buffer_size = int(goal_instruction_buffer_size)
        self._slot: "OrderedDict[str, int]" = OrderedDict()        # text -> row of the table, in insertion order (oldest first)
        self._free: List[int] = []                                 # rows released by evictions / clear
        self._table: Optional[torch.Tensor] = None                 # [capacity, *embedding_shape], allocated at the first insert
        self.buffer_lock = threading.Lock()

    # ------------------------------------------------------------------ reference API
    def get_or_encode_batch(self, texts: Union[str, Sequence[str]]) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        try:
            with self.buffer_lock:
                uncached_texts = [text for text in texts if text not in self._slot]
            if uncached_texts:
                encoded_batch = self.language_encoder(uncached_texts)   # duplicates inside one request are encoded as often as they occur, like the reference
                for text, embedding in zip(uncached_texts, encoded_batch):
                    self.add_to_buffer(text, embedding)
            with self.buffer_lock:
                rows = [self._slot[text] for text in texts]         # KeyError (evicted inside this very request) -> fallback, as in the reference
                idx = torch.tensor(rows, dtype=torch.long, device=self._table.device)
                return self._table.index_select(0, idx)
        except Exception as e:                                       # noqa: BLE001 - the reference catches everything here
            print(f"Error encoding texts: {e}")
            return torch.zeros((len(texts), self.language_encoder.output_dim))

    def add_to_buffer(self, key: str, value: torch.Tensor) -> None:
        with self.buffer_lock:
            value = value.detach()
            if self._table is None:
                self._table = torch.zeros((max(self.goal_instruction_buffer_size, 1),) + tuple(value.shape), dtype=value.dtype, device=value.device)
            if len(self._slot) >= self.goal_instruction_buffer_size and self._slot:
                _, row = self._slot.popitem(last=False)              # FIFO: the oldest insertion goes - also when `key` is already cached
                self._free.append(row)                               # (lang_buffer.py:41-44 pops before it assigns; a duplicate inside one request at capacity evicts twice)
            if key in self._slot:                                    # re-insert of a live key: the value is replaced, its age is not
                self._table[self._slot[key]].copy_(value)
                return
            row = self._free.pop() if self._free else len(self._slot)
            self._table[row].copy_(value)
            self._slot[key] = row

    def get_goal_instruction_embedding(self, goal_instruction: str) -> torch.Tensor:
        return self.get_or_encode_batch([goal_instruction])

    def get_goal_instruction_embeddings(self, goal_instructions: Sequence[str]) -> torch.Tensor:
        return self.get_or_encode_batch(goal_instructions)

    def clear_buffer(self) -> None:
        with self.buffer_lock:
            self._slot.clear()
            self._free = []

    def get_buffer_size(self) -> int:
        with self.buffer_lock:
            return len(self._slot)

    def preload_common_strings(self, goal_instruction_list: Iterable[str]) -> None:
        self.get_or_encode_batch(list(goal_instruction_list))

    def save_buffer(self, filepath: str) -> None:
        """Same file format as the reference: a pickled ``OrderedDict[text -> embedding]`` (oldest first), readable by either side."""
        with self.buffer_lock:
            out = OrderedDict((k, self._table[r].clone()) for k, r in self._slot.items())
            with open(filepath, "wb") as f:
                pickle.dump(out, f)

    def load_buffer(self, filepath: str) -> None:
        with open(filepath, "rb") as f:
            loaded_buffer = pickle.load(f)
        items = list(loaded_buffer.items())[-self.goal_instruction_buffer_size:]
        with self.buffer_lock:
            self._slot.clear()
            self._free = []
            self._table = None
        for k, v in items:
            self.add_to_buffer(k, v)

    # ------------------------------------------------------------------ reference attribute kept for code that peeks at it
    @property
    def goal_instruction_buffer(self) -> "OrderedDict[str, torch.Tensor]":
        with self.buffer_lock:
            return OrderedDict((k, self._table[r]) for k, r in self._slot.items())
