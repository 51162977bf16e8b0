"""Phone-set vocabulary of the task surface (utils/text/text_encoder.py:107-263 in the reference).

Only what the hot path's callers need: the id layout (`<pad>`=0, `<EOS>`=1, `<UNK>`=2 are PREPENDED to the phone list of
`<binary_data_dir>/phone_set.json`, so real phonemes start at id 3 and `len()` = 3 + the non-reserved phones -- that
length sizes `fs.encoder.embed_tokens`), encode / decode, and the silence set the duration losses use
(`is_sil_phoneme`: every token whose first character is not a letter, reserved tokens included).
"""
import json

PAD, EOS, UNK, SEG = "<pad>", "<EOS>", "<UNK>", "|"
RESERVED_TOKENS = [PAD, EOS, UNK]


def is_sil_phoneme(p):
    return p == "" or not p[0].isalpha()


class TokenTextEncoder:
    def __init__(self, vocab_list, replace_oov=UNK):
        toks = RESERVED_TOKENS + [t for t in vocab_list if t not in RESERVED_TOKENS]
        self.id_to_token = dict(enumerate(toks))
        self.token_to_id = {t: i for i, t in self.id_to_token.items()}  # a repeated token keeps its LAST id, as upstream
        self._replace_oov = replace_oov
        self.pad_index, self.eos_index, self.unk_index = (self.token_to_id[t] for t in RESERVED_TOKENS)
        self.seg_index = self.token_to_id.get(SEG, self.eos_index)

    def __len__(self):
        return len(self.id_to_token)

    vocab_size = property(__len__)

    def pad(self):
        return self.pad_index

    def eos(self):
        return self.eos_index

    def unk(self):
        return self.unk_index

    def seg(self):
        return self.seg_index

    def encode(self, s):
        toks = s.strip().split()
        if self._replace_oov is not None:
            toks = [t if t in self.token_to_id else self._replace_oov for t in toks]
        return [self.token_to_id[t] for t in toks]

    def decode(self, ids, strip_eos=False, strip_padding=False):
        ids = [int(i) for i in ids]
        if strip_padding and self.pad_index in ids:
            ids = ids[:ids.index(self.pad_index)]
        if strip_eos and self.eos_index in ids:
            ids = ids[:ids.index(self.eos_index)]
        return " ".join(self.id_to_token.get(i, "ID_%d" % i) for i in ids)

    def sil_phonemes(self):
        return [p for p in self.id_to_token.values() if is_sil_phoneme(p)]

    def sil_ids(self):
        """ids the duration losses treat as silence (speech_editing_base.py:69-72: `encode(p)[0]` for p in sil_phonemes)."""
        return sorted({self.token_to_id[p] for p in self.sil_phonemes() if p != ""})


def build_token_encoder(token_list_file):
    with open(token_list_file) as f:
        return TokenTextEncoder(json.load(f))
