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
    """Attention seq2seq model + post processing network (reference __init__.py:11-97)."""

    def __init__(self, seq2seq, postnet, mel_dim=80, linear_dim=513, n_speakers=1, speaker_embed_dim=16,
                 padding_idx=None, trainable_positional_encodings=False,
                 use_decoder_state_for_postnet_input=False, speaker_embedding_weight_std=0.01,
                 freeze_embedding=False):
        super(MultiSpeakerTTSModel, self).__init__()
        self.seq2seq = seq2seq
        self.postnet = postnet  # referred as "Converter" in DeepVoice3
        self.mel_dim = mel_dim
        self.linear_dim = linear_dim
        self.trainable_positional_encodings = trainable_positional_encodings
        self.use_decoder_state_for_postnet_input = use_decoder_state_for_postnet_input
        self.freeze_embedding = freeze_embedding
        if trainable_positional_encodings:
            raise NotImplementedError("trainable_positional_encodings=True is not supported by the HIP path "
                                      "(no preset of the reference enables it)")
        if n_speakers > 1:
            self.embed_speakers = Embedding(n_speakers, speaker_embed_dim, padding_idx=None,
                                            std=speaker_embedding_weight_std)
        self.n_speakers = n_speakers
        self.speaker_embed_dim = speaker_embed_dim
        from .deepvoice3 import _name_sites
        _name_sites(self, "model")

    def make_generation_fast_(self):
        """Fold weight norm back into plain weights (reference __init__.py:39-46)."""
        for m in self.modules():
            if isinstance(m, _conv._WNLayer):
                try:
                    m.remove_weight_norm_()
                except ValueError:  # this module didn't have weight norm
                    pass

    def get_trainable_parameters(self):
        freezed_param_ids = set()
        encoder, decoder = self.seq2seq.encoder, self.seq2seq.decoder
        if not self.trainable_positional_encodings:
            pe_query_param_ids = set(map(id, decoder.embed_query_positions.parameters()))
            pe_keys_param_ids = set(map(id, decoder.embed_keys_positions.parameters()))
            freezed_param_ids |= (pe_query_param_ids | pe_keys_param_ids)
        if self.freeze_embedding:
            embed_param_ids = set(map(id, encoder.embed_tokens.parameters()))
            freezed_param_ids |= embed_param_ids
        return (p for p in self.parameters() if id(p) not in freezed_param_ids)

    def forward(self, text_sequences, mel_targets=None, speaker_ids=None, text_positions=None,
                frame_positions=None, input_lengths=None):
        B = text_sequences.size(0)
        if speaker_ids is not None:
            assert self.n_speakers > 1
            speaker_embed = self.embed_speakers(speaker_ids)
        else:
            speaker_embed = None

        # (B, T//r, mel_dim*r)
        mel_outputs, alignments, done, decoder_states = self.seq2seq(
            text_sequences, mel_targets, speaker_embed, text_positions, frame_positions, input_lengths)

        # (B, T, mel_dim)   [reshape: identical values to the reference's .view on old torch]
        mel_outputs = mel_outputs.reshape(B, -1, self.mel_dim)

        if self.use_decoder_state_for_postnet_input:
            T = mel_outputs.size(1)
            postnet_inputs = decoder_states.view(B, T, -1)
            if postnet_inputs.shape == decoder_states.shape:
                # r == 1: hand the BCT image the decoder already holds to the converter
                bct = getattr(decoder_states, "_dv3_bct", None)
                if bct is not None:
                    postnet_inputs._dv3_bct = bct
        else:
            postnet_inputs = mel_outputs

        linear_outputs = self.postnet(postnet_inputs, speaker_embed)
        assert linear_outputs.size(-1) == self.linear_dim
        return mel_outputs, linear_outputs, alignments, done


class AttentionSeq2Seq(nn.Module):
    """Encoder + Decoder with attention (reference __init__.py:100-126)."""

    def __init__(self, encoder, decoder):
        super(AttentionSeq2Seq, self).__init__()
        self.encoder = encoder
        self.decoder = decoder
        if isinstance(self.decoder.attention, nn.ModuleList):
            self.encoder.num_attention_layers = sum([layer is not None for layer in decoder.attention])

    def forward(self, text_sequences, mel_targets=None, speaker_embed=None, text_positions=None,
                frame_positions=None, input_lengths=None):
        encoder_outputs = self.encoder(text_sequences, lengths=input_lengths, speaker_embed=speaker_embed)
        mel_outputs, alignments, done, decoder_states = self.decoder(
            encoder_outputs, mel_targets, text_positions=text_positions, frame_positions=frame_positions,
            speaker_embed=speaker_embed, lengths=input_lengths)
        return mel_outputs, alignments, done, decoder_states
