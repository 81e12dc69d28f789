"""TextProcessor -- host-side text plumbing of the hot path, behaviour-identical to the reference
(vietvoicetts/core/text_processor.py): vocab file -> {char: line number} (:19-28), char -> int32 ids
with unknown -> 0 (:30-37), UTF-8 byte length + 3 per pause match (:39-41), text cleaning (:43-74),
sentence / comma / word chunking with short-chunk merging (:76-175).  Pinned by the golden vectors
in tests/golden/host_golden.json (generated from the reference functions themselves).

Reference quirk kept on purpose: ``pause_punc`` is used as a REGEX, so the default r".,?!:" almost
never matches.
"""
from __future__ import annotations

import logging
import re
from pathlib import Path
from typing import Dict, List

import numpy as np

logger = logging.getLogger("vietvoicetts")

_ASCII = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
_VIET = "àáảãạăằắẳẵặâầấẩẫậèéẻẽẹêềếểễệđìíỉĩịòóỏõọôồốổỗộơờớởỡợùúủũụưừứửữựỳỵỷỹýỳỵỷỹ"
_PUNCT = " .,!?'@$%&/:;()"
_KEEP = frozenset(_ASCII + _ASCII.upper() + _VIET + _VIET.upper() + _PUNCT)
_SENTENCE_END = re.compile(r"(?<=[.!?]) +")


class TextProcessor:
    def __init__(self, vocab_path: str):
        self.vocab_char_map = self._load_vocab(vocab_path)
        self.vocab_size = len(self.vocab_char_map)

    def _load_vocab(self, vocab_path: str) -> Dict[str, int]:
        if not Path(vocab_path).exists():
            raise FileNotFoundError(f"Vocabulary file not found: {vocab_path}")
        table: Dict[str, int] = {}
        with open(vocab_path, "r", encoding="utf-8") as fh:
            for line_no, line in enumerate(fh):
                table[line.rstrip("\n")] = line_no      # later duplicates win, like a dict assignment
        return table

    def text_to_indices(self, texts: List[List[str]]) -> np.ndarray:
        lookup = self.vocab_char_map.get
        rows = [np.fromiter((lookup(ch, 0) for ch in chars), dtype=np.int32, count=len(chars)) for chars in texts]
        return np.stack(rows, axis=0)

    def calculate_text_length(self, text: str, pause_punc: str) -> int:
        return len(text.encode("utf-8")) + 3 * len(re.findall(pause_punc, text))

    def clean_text(self, text: str) -> str:
        if "\n" in text:
            lines = [ln.strip() for ln in text.split("\n")]
            lines = [ln if ln.endswith(".") else ln + "." for ln in lines if ln]
            text = " ".join(lines)
        text = "".join(ch if ch in _KEEP else " " for ch in text).strip()
        text = re.sub(r"[;:()]", ",", text)
        text = re.sub(r"\.+", ".", text)
        text = re.sub(r",+", ",", text)
        text = re.sub(r"\s+", " ", text)
        if not text.endswith((".", "?", "!", ",")):
            text += "."
        return text

    # ------------------------------------------------------------------ chunking
    @staticmethod
    def _split_words(part: str, max_chars: int) -> List[str]:
        pieces, cur = [], ""
        for word in part.split():
            if cur and len(cur) + 1 + len(word) > max_chars:
                pieces.append(cur)
                cur = word
            else:
                cur = f"{cur} {word}" if cur else word
        if cur.strip():
            pieces.append(cur.strip())
        return pieces

    @classmethod
    def _pieces(cls, text: str, max_chars: int) -> List[str]:
        out: List[str] = []
        for sentence in _SENTENCE_END.split(text.strip()):
            sentence = sentence.strip()
            if not sentence:
                continue
            if len(sentence) <= max_chars:
                out.append(sentence)
                continue
            for part in sentence.split(", "):
                part = part.strip()
                if not part:
                    continue
                if len(part) <= max_chars:
                    out.append(part)
                else:
                    logger.warning("Part too long (%d chars), splitting at word boundaries: %s...", len(part), part[:50])
                    out.extend(cls._split_words(part, max_chars))
        return out

    def chunk_text(self, text: str, max_chars: int = 135) -> List[str]:
        if not text.strip():
            return []
        pieces = self._pieces(text, max_chars)
        if not pieces:
            return []
        # greedy packing of pieces into chunks of at most max_chars
        packed, cur = [], ""
        for piece in pieces:
            if cur and len(cur) + 1 + len(piece) > max_chars:
                packed.append(cur.strip())
                cur = piece
            else:
                cur = f"{cur} {piece}" if cur else piece
        if cur:
            packed.append(cur.strip())
        # merge chunks of fewer than 4 words into a neighbour when the result still fits
        merged: List[str] = []
        i, n = 0, len(packed)
        while i < n:
            chunk = packed[i]
            if n > 1 and len(chunk.split()) < 4:
                if i < n - 1:
                    joined = f"{chunk} {packed[i + 1]}"
                    if len(joined) <= max_chars:
                        merged.append(joined)
                        i += 2
                        continue
                elif merged:
                    joined = f"{merged[-1]} {chunk}"
                    if len(joined) <= max_chars:
                        merged[-1] = joined
                        i += 1
                        continue
            merged.append(chunk)
            i += 1
        logger.debug("chunk_text: %d chunks %s (max %d)", len(merged), [len(c) for c in merged], max_chars)
        return merged
