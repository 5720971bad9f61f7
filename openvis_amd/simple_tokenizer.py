"""CLIP byte-level BPE tokenizer (host side of A11).

Behaviour of third_parties/mask_adapted_clip/mask_adapted_clip/simple_tokenizer.py:68-150 + clip.py:239-283 (`tokenize`):
lower-cased, whitespace-collapsed text -> regex word split -> bytes mapped to printable unicode -> greedy lowest-rank
BPE merges -> ids; sequences are `<|startoftext|> ids <|endoftext|>` zero-padded to 77.

The merge table is OpenAI CLIP's public `bpe_simple_vocab_16e6.txt.gz`; it is NOT shipped here.  Pass its path, set
`OVIS_CLIP_BPE`, or install it next to a `clip` / `mask_adapted_clip` package.  `ftfy.fix_text` (mojibake repair) is
applied when ftfy is importable and skipped otherwise (identity on clean text such as dataset class names)."""
import gzip
import html
import os

import regex

_SPLIT = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                       regex.IGNORECASE)
CONTEXT_LENGTH = 77
N_MERGES = 49152 - 256 - 2


def _byte_alphabet():
    """256 bytes -> printable unicode characters (printable latin-1 kept, the rest moved to U+0100...)."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table, extra = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table, keep + [b for b in range(256) if b not in keep]


def find_bpe_file(path=None):
    cands = [path, os.environ.get("OVIS_CLIP_BPE")]
    for mod in ("clip", "mask_adapted_clip"):
        try:
            m = __import__(mod)
            cands.append(os.path.join(os.path.dirname(m.__file__), "bpe_simple_vocab_16e6.txt.gz"))
        except Exception:
            pass
    for c in cands:
        if c and os.path.isfile(c):
            return c
    raise FileNotFoundError("CLIP BPE merge table not found: pass bpe_path, or set OVIS_CLIP_BPE to bpe_simple_vocab_16e6.txt.gz")


class SimpleTokenizer:
    def __init__(self, bpe_path=None):
        table, order = _byte_alphabet()
        self.byte_char = table
        lines = gzip.open(find_bpe_file(bpe_path)).read().decode("utf-8").split("\n")
        merges = [tuple(l.split()) for l in lines[1:N_MERGES + 1]]
        symbols = [table[b] for b in order]
        vocab = symbols + [s + "</w>" for s in symbols] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        self._cache = {"<|startoftext|>": ["<|startoftext|>"], "<|endoftext|>": ["<|endoftext|>"]}

    def _merge_word(self, token):
        """token (already byte-mapped) -> list of BPE symbols: repeatedly fuse the adjacent pair of lowest merge rank."""
        if token in self._cache:
            return self._cache[token]
        parts = list(token[:-1]) + [token[-1] + "</w>"]
        while len(parts) > 1:
            best, best_rank = None, None
            for a, b in zip(parts, parts[1:]):
                r = self.rank.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            fused, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    fused.append(best[0] + best[1])
                    i += 2
                else:
                    fused.append(parts[i])
                    i += 1
            parts = fused
        self._cache[token] = parts
        return parts

    @staticmethod
    def _clean(text):
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text)).strip()
        return regex.sub(r"\s+", " ", text).strip().lower()

    def encode(self, text):
        ids = []
        for word in regex.findall(_SPLIT, self._clean(text)):
            mapped = "".join(self.byte_char[b] for b in word.encode("utf-8"))
            ids.extend(self.encoder[s] for s in self._merge_word(mapped))
        return ids

    def tokenize(self, texts, context_length=CONTEXT_LENGTH):
        """list[str] -> int64 tensor [len(texts), context_length] (clip.py:263-283; too-long input raises)."""
        import torch
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            out[i, :len(ids)] = torch.tensor(ids)
        return out
