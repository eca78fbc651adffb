# coding: utf-8
"""deepvoice3_pytorch_amd -- the DeepVoice3 / Nyanko hot path on hand-written MI355X (gfx950) HIP
kernels, behind the module API of r9y9/deepvoice3_pytorch.

Drop-in surface (reference deepvoice3_pytorch/__init__.py:11-126, builder.py:7-258):
    from deepvoice3_pytorch_amd import builder
    model = builder.deepvoice3(n_vocab=..., ...)      # / nyanko / deepvoice3_multispeaker
    mel, linear, alignments, done = model(text, mel_targets, speaker_ids, text_positions,
                                          frame_positions, input_lengths)
    model.load_state_dict(reference_checkpoint["state_dict"])   # identical keys and shapes

Nothing here falls back to torch CPU/GPU ops for the conv stacks, attention, losses or the
optimiser tail: if libdv3hip.so is missing the first op raises.
"""
__version__ = "0.1.0"

import torch
from torch import nn

from .modules import Embedding
from . import conv as _conv


class MultiSpeakerTTSModel(nn.Module):
    """seq2seq (Encoder + attention Decoder) followed by the post-net ("Converter"), with an optional
    speaker-embedding table -- the object the reference's builders return
    (deepvoice3_pytorch/__init__.py:11-97): same constructor keywords, attributes and call signature."""

    def __init__(self, seq2seq, postnet, mel_dim=80, linear_dim=513, n_speakers=1, speaker_embed_dim=16,
                 padding_idx=None, trainable_positional_encodings=False,
                 use_decoder_state_for_postnet_input=False, speaker_embedding_weight_std=0.01,
                 freeze_embedding=False):
        nn.Module.__init__(self)
        self.seq2seq, self.postnet = seq2seq, postnet
        self.mel_dim, self.linear_dim = mel_dim, linear_dim
        self.n_speakers, self.speaker_embed_dim = n_speakers, speaker_embed_dim
        self.trainable_positional_encodings = trainable_positional_encodings
        self.use_decoder_state_for_postnet_input = use_decoder_state_for_postnet_input
        self.freeze_embedding = freeze_embedding
        if n_speakers > 1:     # state_dict key "embed_speakers.weight", N(0, std) init like the reference
            self.embed_speakers = Embedding(n_speakers, speaker_embed_dim, padding_idx=None,
                                            std=speaker_embedding_weight_std)
        from .deepvoice3 import _name_sites
        _name_sites(self, "model")     # stable dropout-site names for the Philox streams

    def make_generation_fast_(self):
        """Inference: replace every weight_g / weight_v pair by the plain weight (__init__.py:39-46)."""
        for layer in (m for m in self.modules() if isinstance(m, _conv._WNLayer)):
            try:
                layer.remove_weight_norm_()
            except ValueError:
                continue           # plain (never weight-normed) layer

    def get_trainable_parameters(self):
        """Everything except the frozen sinusoid tables (and the text embedding when
        freeze_embedding is set), as a generator -- reference __init__.py:48-63."""
        dec, enc = self.seq2seq.decoder, self.seq2seq.encoder
        frozen = [] if self.trainable_positional_encodings else [dec.embed_query_positions, dec.embed_keys_positions]
        if self.freeze_embedding:
            frozen.append(enc.embed_tokens)
        skip = {id(p) for mod in frozen for p in mod.parameters()}
        return (p for p in self.parameters() if id(p) not in skip)

    def forward(self, text_sequences, mel_targets=None, speaker_ids=None, text_positions=None,
                frame_positions=None, input_lengths=None):
        speaker_embed = None
        if speaker_ids is not None:
            assert self.n_speakers > 1
            speaker_embed = self.embed_speakers(speaker_ids)
        mel, alignments, done, states = self.seq2seq(text_sequences, mel_targets, speaker_embed,
                                                     text_positions, frame_positions, input_lengths)
        n = text_sequences.size(0)
        mel = mel.reshape(n, -1, self.mel_dim)            # (B, T//r, mel_dim*r) -> (B, T, mel_dim)
        if not self.use_decoder_state_for_postnet_input:
            post_in = mel
        else:
            post_in = states.view(n, mel.size(1), -1)
            bct = getattr(states, "_dv3_bct", None)
            if bct is not None and post_in.shape == states.shape:
                post_in._dv3_bct = bct                    # r == 1: reuse the decoder's BCT image
        linear = self.postnet(post_in, speaker_embed)
        assert linear.size(-1) == self.linear_dim
        return mel, linear, alignments, done


class AttentionSeq2Seq(nn.Module):
    """Encoder -> attention Decoder pair (reference __init__.py:100-126)."""

    def __init__(self, encoder, decoder):
        nn.Module.__init__(self)
        self.encoder, self.decoder = encoder, decoder
        att = getattr(decoder, "attention", None)
        if isinstance(att, nn.ModuleList):
            encoder.num_attention_layers = sum(1 for layer in att if layer is not None)

    def forward(self, text_sequences, mel_targets=None, speaker_embed=None, text_positions=None,
                frame_positions=None, input_lengths=None):
        memory = self.encoder(text_sequences, lengths=input_lengths, speaker_embed=speaker_embed)
        return self.decoder(memory, mel_targets, text_positions=text_positions,
                            frame_positions=frame_positions, speaker_embed=speaker_embed,
                            lengths=input_lengths)
