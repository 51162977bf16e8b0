"""CampNet masked-mel transformer (BASELINE config 5).

Module tree and `state_dict` keys follow modules/speech_editing/campnet/campnet.py:14-41 and
modules/speech_editing/commons/transformer.py (EncSALayer :489-528, DecSALayer :531-609, TransformerEncoder :712-747,
TransformerDecoder :750-811), so a reference checkpoint loads with strict=True -- including the members the reference
inherits from modules/tts/fs.py FastSpeech and never runs (`mel_out`, `pitch_embed`, `pitch_predictor`,
`encoder.pre_net`).  Everything runs on the [B][C][T] layout of the rest of the library: Linear = 1x1 conv, LayerNorm
over channels, heads = channel slices addressed through strides (no transposes), attention = one fused kernel per layer and
direction (QK^T on MFMA -> masked fp32 online softmax in registers -> PV, scores never in HBM; csrc/attention_fused.hip).  With an autograd tape every node (forward and backward) is a kernel of libset_amd.so.
egs/campnet.yaml sets every dropout to 0, so there is no train/eval difference in the arithmetic.
"""
import math

import torch
from torch import nn

from . import ops
from .fs import ConvBlocks, MelEncoder, PitchPredictor, _backend, _embedding

MAX_POSITIONS = 2000  # DEFAULT_MAX_TARGET_POSITIONS, transformer.py:11


def sinusoid_table(n, dim, padding_idx=0):
    """transformer.py:31-48, built once on the host with the same fp32 torch ops as the reference builds its own."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(n, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(n, -1)
    if dim % 2 == 1:
        e = torch.cat([e, torch.zeros(n, 1)], dim=1)
    e[padding_idx, :] = 0
    return e


class _Positions(nn.Module):
    """SinusoidalPositionalEmbedding (transformer.py:14-73): only `_float_tensor` is in the state_dict; the table is a
    plain attribute there and a non-persistent buffer here."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.register_buffer("_float_tensor", torch.zeros(1))
        self.register_buffer("table", sinusoid_table(MAX_POSITIONS, dim), persistent=False)

    def table_for(self, T):
        if T + 1 > self.table.shape[0]:
            self.table = sinusoid_table(T + 1, self.dim).to(self.table.device)
        return self.table


class MultiheadAttention(nn.Module):
    """Bias-free packed projections (transformer.py:138-189 with bias=False, qkv_same_dim)."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.scaling = (embed_dim // num_heads) ** -0.5
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=False)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.xavier_uniform_(self.out_proj.weight)
        H = embed_dim
        self._w_qkv = ops.ConvWeight((self, "in_proj_weight"), 3 * H, H, 1)
        self._w_q = ops.ConvWeight((self, "in_proj_weight"), H, H, 1)
        self._w_kv = ops.ConvWeight((self, "in_proj_weight"), 2 * H, H, 1, base=H * H)
        self._w_out = ops.ConvWeight((self, "out_proj.weight"), H, H, 1)

    def self_attn(self, h, res, key_padding=None, mask=None):
        """(res + out_proj(attention(h))) (* mask).  torch's multi_head_attention_forward path: -inf padding fill."""
        F = _backend()
        qkv = F.conv1d(h, self._w_qkv)
        o, _ = F.self_attention(qkv, self.num_heads, key_padding, float("-inf"), self.scaling)
        return F.conv1d(o, self._w_out, res=res, mask=mask)

    def cross_attn(self, h, enc, res, enc_padding, want_p=True):
        """encoder-decoder attention, the module's own path (static_kv=True, transformer.py:283-410): -1e8 fill.
        Returns (res + out_proj(...), probabilities [B, heads, T, T_txt] -- only when asked: the fused kernel keeps them
        on chip and the decoder returns the first layer's alone)."""
        F = _backend()
        q = F.conv1d(h, self._w_q)
        kv = F.conv1d(enc, self._w_kv)
        o, p = F.cross_attention(q, kv, self.num_heads, enc_padding, -1e8, self.scaling, want_p=want_p)
        return F.conv1d(o, self._w_out, res=res), p


class TransformerFFNLayer(nn.Module):
    """conv k (H -> 4H; 'SAME' or causal 'LEFT' padding) * k^-1/2 -> GELU -> Linear (transformer.py:76-113)."""

    def __init__(self, hidden_size, filter_size, padding="SAME", kernel_size=1):
        super().__init__()
        self.kernel_size, self.padding = kernel_size, padding
        conv = nn.Conv1d(hidden_size, filter_size, kernel_size, padding=kernel_size // 2 if padding == "SAME" else 0)
        # 'LEFT' is Sequential(ConstantPad1d, Conv1d) upstream: the conv's parameters live under `ffn_1.1.*`
        self.ffn_1 = conv if padding == "SAME" else nn.Sequential(nn.Identity(), conv)
        self.ffn_2 = nn.Linear(filter_size, hidden_size)
        object.__setattr__(self, "_conv", conv)  # plain reference: the parameters are registered once, under ffn_1
        self._w1 = ops.ConvWeight((self, "_conv.weight"), filter_size, hidden_size, kernel_size)
        self._w2 = ops.ConvWeight((self, "ffn_2.weight"), hidden_size, filter_size, 1)

    def run(self, h, res, mask=None):
        F = _backend()
        k = self.kernel_size
        pad = k // 2 if self.padding == "SAME" else k - 1  # 'LEFT': k-1 zeros in front, output length = input length
        f = F.conv1d(h, self._w1, self._conv.bias, pad=pad, alpha=k ** -0.5, act="gelu", T_out=h.shape[2])
        return F.conv1d(f, self._w2, self.ffn_2.bias, res=res, mask=mask)

    def run_preln(self, x, ln, mask=None):
        """x + FFN(LayerNorm(x)) (* mask): the sub-block of Enc/DecSALayer as ONE tape node in training (autograd_ops._PreLnFfnFn)."""
        k = self.kernel_size
        pad = k // 2 if self.padding == "SAME" else k - 1
        return _backend().preln_ffn(x, (ln.weight, ln.bias), self._w1, self._conv.bias, self._w2, self.ffn_2.bias, pad=pad, alpha=k ** -0.5,
                                    act="gelu", mask=mask, eps=ln.eps, T_out=x.shape[2])


class EncSALayer(nn.Module):
    def __init__(self, c, num_heads, kernel_size=9):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(c)
        self.self_attn = MultiheadAttention(c, num_heads)
        self.layer_norm2 = nn.LayerNorm(c)
        self.ffn = TransformerFFNLayer(c, 4 * c, kernel_size=kernel_size, padding="SAME")

    def run(self, x, key_padding, keep):
        F = _backend()
        ln = self.layer_norm1
        x = F.preln_self_attn(x, (ln.weight, ln.bias), self.self_attn, key_padding, keep, eps=ln.eps)  # one tape node in training
        return self.ffn.run_preln(x, self.layer_norm2, keep)


class DecSALayer(nn.Module):
    def __init__(self, c, num_heads, kernel_size=9):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(c)
        self.self_attn = MultiheadAttention(c, num_heads)
        self.layer_norm2 = nn.LayerNorm(c)
        self.encoder_attn = MultiheadAttention(c, num_heads)
        self.layer_norm3 = nn.LayerNorm(c)
        self.ffn = TransformerFFNLayer(c, 4 * c, padding="LEFT", kernel_size=kernel_size)

    def run(self, x, enc, enc_padding, keep, want_p=True):
        """The self-attention gets NO padding mask (the reference calls the layer without self_attn_padding_mask,
        transformer.py:803); returns (x, cross-attention probabilities)."""
        F = _backend()
        ln1, ln2 = self.layer_norm1, self.layer_norm2
        x = F.preln_self_attn(x, (ln1.weight, ln1.bias), self.self_attn, eps=ln1.eps)
        x, p = F.preln_cross_attn(x, (ln2.weight, ln2.bias), self.encoder_attn, enc, enc_padding, want_p, eps=ln2.eps)
        return self.ffn.run_preln(x, self.layer_norm3, keep), p


class _Layer(nn.Module):
    """TransformerEncoderLayer / TransformerDecoderLayer: the work is in `.op` (transformer.py:619-652)."""

    def __init__(self, op):
        super().__init__()
        self.op = op


def _one_minus(m):
    return ops.mul_one_minus_mask(torch.ones_like(m), m, 1)


class TransformerEncoder(nn.Module):
    """transformer.py:712-747 on FFTBlocks :655-709 (use_pos_embed=False there; its own embed_positions here)."""

    def __init__(self, dict_size, hidden_size, num_layers=3, kernel_size=9, num_heads=2):
        super().__init__()
        self.hidden_size = hidden_size
        self.layers = nn.ModuleList([_Layer(EncSALayer(hidden_size, num_heads, kernel_size)) for _ in range(num_layers)])
        self.layer_norm = nn.LayerNorm(hidden_size)
        self.embed_tokens = _embedding(dict_size, hidden_size, 0)
        self.pre_net = ConvBlocks(hidden_size, hidden_size, [1] * 3, 1, layers_in_block=2)  # never run upstream (:743)
        self.embed_scale = math.sqrt(hidden_size)
        self.embed_positions = _Positions(hidden_size)

    def run(self, txt_tokens):
        F = _backend()
        keep = ops.index_mask(txt_tokens)
        pad = _one_minus(keep)
        x = F.embedding_bct(txt_tokens, self.embed_tokens.weight, scale=self.embed_scale, padding_idx=0)
        pos = ops.make_positions(tokens=txt_tokens)
        x = F.embedding_bct(pos, self.embed_positions.table_for(txt_tokens.shape[1]), out=x, accumulate=True)
        x = F.add_chan_mask(x, None, keep)
        for layer in self.layers:
            x = layer.op.run(x, pad, keep)
        return F.layernorm_ch(x, self.layer_norm.weight, self.layer_norm.bias, mask=keep), keep


class TransformerDecoder(nn.Module):
    """transformer.py:750-811: the coarse decoder over the masked mel frames."""

    def __init__(self, hidden_size, num_layers=6, ffn_kernel_size=9, num_heads=2):
        super().__init__()
        self.pos_embed_alpha = nn.Parameter(torch.ones(1))
        self.embed_positions = _Positions(hidden_size)
        self.layers = nn.ModuleList([_Layer(DecSALayer(hidden_size, num_heads, ffn_kernel_size))
                                     for _ in range(num_layers)])
        self.layer_norm = nn.LayerNorm(hidden_size)

    def run(self, x, enc):
        """x [B,H,T], enc [B,H,T_txt] -> (x, attn [B,T,T_txt] = head mean of the FIRST layer's probabilities)."""
        F = _backend()
        enc_pad = _one_minus(ops.abs_sum_mask(enc.detach()))
        keep = ops.abs_sum_mask(x.detach())
        pos = ops.make_positions(x_bct=x.detach())  # numbered by channel 0 != 0, like `x[..., 0]` upstream (:795)
        x = F.pos_add(x, self.pos_embed_alpha, pos, self.embed_positions.table_for(x.shape[2]))
        x = F.add_chan_mask(x, None, keep)
        encs = F.fanout(enc, len(self.layers))
        attn = None
        for layer, e in zip(self.layers, encs):
            x, p = layer.op.run(x, e, enc_pad, keep, want_p=attn is None)
            if attn is None:
                attn = ops.head_mean(p.detach())
        return F.layernorm_ch(x, self.layer_norm.weight, self.layer_norm.bias, mask=keep), attn


class CampNet(nn.Module):
    """modules/speech_editing/campnet/campnet.py:14-69."""

    def __init__(self, ph_dict_size, word_dict_size, hparams, out_dims=None):
        super().__init__()
        hp = self.hparams = dict(hparams)
        H = self.hidden_size = hp["hidden_size"]
        self.out_dims = hp["audio_num_mel_bins"] if out_dims is None else out_dims
        k = hp["dec_ffn_kernel_size"]
        self.encoder = TransformerEncoder(ph_dict_size, H, num_layers=3, kernel_size=k, num_heads=2)
        # inherited from modules/tts/fs.py FastSpeech.__init__ and kept by CampNet (never run): checkpoint keys only
        self.mel_out = nn.Linear(H, self.out_dims, bias=True)
        if hp.get("use_pitch_embed", True):
            ph = hp["predictor_hidden"] if hp["predictor_hidden"] > 0 else H
            self.pitch_embed = _embedding(300, H, 0)
            self.pitch_predictor = PitchPredictor(H, n_chans=ph, n_layers=5, dropout_rate=0.1, odim=2,
                                                  kernel_size=hp["predictor_kernel"])
        self.mel_encoder = MelEncoder(hidden_size=H)
        self.decoder_coarse = TransformerDecoder(H, num_layers=6, ffn_kernel_size=k, num_heads=2)
        self.decoder_fine = ConvBlocks(H, H, [1] * 5, 5, layers_in_block=2)
        self.mel_out_coarse = nn.Linear(H, self.out_dims, bias=False)
        self.mel_out_fine = nn.Linear(H, self.out_dims, bias=False)
        self.mask_emb = nn.Parameter(torch.zeros(1, 1, 80))
        self._w_coarse = ops.ConvWeight((self, "mel_out_coarse.weight"), self.out_dims, H, 1)
        self._w_fine = ops.ConvWeight((self, "mel_out_fine.weight"), self.out_dims, H, 1)

    def forward(self, txt_tokens, spk_embed=None, spk_id=None, mels=None, stutter_mel_masks=None, time_mel_masks=None,
                infer=False, global_step=None, *args, **kwargs):
        """Returns the reference's keys `mel_out_coarse`, `mel_out_fine` ([B,T,80]), `attn` ([B,T,T_txt]) plus the
        internal-layout tensors `*_bct` the task computes its losses from."""
        if not torch.cuda.is_available():
            raise RuntimeError("CampNet (set_amd) needs an MI355X: there is no CPU fallback for this path")
        F = _backend()
        B, T, M = mels.shape
        tm = time_mel_masks.reshape(B, T).contiguous()
        enc, src_keep = self.encoder.run(txt_tokens)  # already * src_nonpadding (campnet.py:51-52 is idempotent)
        mels_bct = ops.btc_to_bct(mels.contiguous())
        mel_keep = ops.abs_sum_mask(mels_bct)
        # coarse decoder
        x = F.mask_fill_chan(mels_bct, self.mask_emb, tm)                      # mels*(1-mask) + mask_emb*mask
        x = self.mel_encoder.run(x, mask=mel_keep)
        h, attn = self.decoder_coarse.run(x, enc)
        h = F.add_chan_mask(h, None, mel_keep)
        coarse = F.conv1d(h, self._w_coarse, mask=mel_keep)                    # [B,80,T]
        coarse_out, coarse_in = F.fanout(coarse, 2)
        # fine decoder on the coarse paste (gradients flow through it, the reference does not detach)
        base = ops.add_chan_mask(mels_bct, None, _one_minus(tm))               # mels*(1-mask), no gradient
        mel_coarse = F.add_masked(base, coarse_in, tm)
        mc_enc, mc_res = F.fanout(mel_coarse, 2)
        x = self.mel_encoder.run(mc_enc, mask=mel_keep)
        f = self.decoder_fine.run(x)
        f = F.add_chan_mask(f, None, mel_keep)
        f = F.conv1d(f, self._w_fine, mask=mel_keep)
        fine = F.add_masked(mc_res, f, tm)
        ret = {"mel_out_coarse_bct": coarse_out, "mel_out_fine_bct": fine, "attn": attn}
        ret["mel_out_coarse"] = ops.bct_to_btc(coarse_out.detach())
        ret["mel_out_fine"] = ops.bct_to_btc(fine.detach())
        return ret
