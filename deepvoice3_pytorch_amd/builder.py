# coding: utf-8
"""Model builders -- same names, keyword arguments and defaults as the reference's
deepvoice3_pytorch/builder.py:7-258; they return a MultiSpeakerTTSModel whose modules run on
the HIP kernels of this package."""
from . import MultiSpeakerTTSModel, AttentionSeq2Seq


def _dv3(n_vocab, preattention_dils, attention, embed_dim, mel_dim, linear_dim, r, downsample_step,
         n_speakers, speaker_embed_dim, padding_idx, dropout, kernel_size, encoder_channels,
         decoder_channels, converter_channels, query_position_rate, key_position_rate, use_memory_mask,
         trainable_positional_encodings, force_monotonic_attention, use_decoder_state_for_postnet_input,
         max_positions, embedding_weight_std, speaker_embedding_weight_std, freeze_embedding,
         window_ahead, window_backward, key_projection, value_projection):
    from .deepvoice3 import Encoder, Decoder, Converter

    time_upsampling = max(downsample_step // r, 1)
    h, k = encoder_channels, kernel_size
    encoder = Encoder(
        n_vocab, embed_dim, padding_idx=padding_idx, n_speakers=n_speakers,
        speaker_embed_dim=speaker_embed_dim, dropout=dropout, max_positions=max_positions,
        embedding_weight_std=embedding_weight_std,
        convolutions=[(h, k, d) for d in (1, 3, 9, 27, 1, 3, 9, 27, 1, 3)])

    h = decoder_channels
    decoder = Decoder(
        embed_dim, in_dim=mel_dim, r=r, padding_idx=padding_idx, n_speakers=n_speakers,
        speaker_embed_dim=speaker_embed_dim, dropout=dropout, max_positions=max_positions,
        preattention=[(h, k, d) for d in preattention_dils],
        convolutions=[(h, k, d) for d in (1, 3, 9, 27, 1)],
        attention=attention, force_monotonic_attention=force_monotonic_attention,
        query_position_rate=query_position_rate, key_position_rate=key_position_rate,
        use_memory_mask=use_memory_mask, window_ahead=window_ahead, window_backward=window_backward,
        key_projection=key_projection, value_projection=value_projection)

    seq2seq = AttentionSeq2Seq(encoder, decoder)

    in_dim = h // r if use_decoder_state_for_postnet_input else mel_dim
    h = converter_channels
    converter = Converter(
        n_speakers=n_speakers, speaker_embed_dim=speaker_embed_dim, in_dim=in_dim, out_dim=linear_dim,
        dropout=dropout, time_upsampling=time_upsampling,
        convolutions=[(h, k, 1), (h, k, 3), (2 * h, k, 1), (2 * h, k, 3)])

    return MultiSpeakerTTSModel(
        seq2seq, converter, padding_idx=padding_idx, mel_dim=mel_dim, linear_dim=linear_dim,
        n_speakers=n_speakers, speaker_embed_dim=speaker_embed_dim,
        trainable_positional_encodings=trainable_positional_encodings,
        use_decoder_state_for_postnet_input=use_decoder_state_for_postnet_input,
        speaker_embedding_weight_std=speaker_embedding_weight_std, freeze_embedding=freeze_embedding)


def deepvoice3(n_vocab, embed_dim=256, mel_dim=80, linear_dim=513, r=4, downsample_step=1, n_speakers=1,
               speaker_embed_dim=16, padding_idx=0, dropout=(1 - 0.95), kernel_size=5,
               encoder_channels=128, decoder_channels=256, converter_channels=256,
               query_position_rate=1.0, key_position_rate=1.29, use_memory_mask=False,
               trainable_positional_encodings=False, force_monotonic_attention=True,
               use_decoder_state_for_postnet_input=True, max_positions=512, embedding_weight_std=0.1,
               speaker_embedding_weight_std=0.01, freeze_embedding=False, window_ahead=3,
               window_backward=1, key_projection=False, value_projection=False):
    """Build deepvoice3 (reference builder.py:7-93)."""
    return _dv3(n_vocab, (1, 3), [True, False, False, False, True], embed_dim, mel_dim, linear_dim, r,
                downsample_step, n_speakers, speaker_embed_dim, padding_idx, dropout, kernel_size,
                encoder_channels, decoder_channels, converter_channels, query_position_rate,
                key_position_rate, use_memory_mask, trainable_positional_encodings,
                force_monotonic_attention, use_decoder_state_for_postnet_input, max_positions,
                embedding_weight_std, speaker_embedding_weight_std, freeze_embedding, window_ahead,
                window_backward, key_projection, value_projection)


def deepvoice3_multispeaker(n_vocab, embed_dim=256, mel_dim=80, linear_dim=513, r=4, downsample_step=1,
                            n_speakers=1, speaker_embed_dim=16, padding_idx=0, dropout=(1 - 0.95),
                            kernel_size=5, encoder_channels=128, decoder_channels=256,
                            converter_channels=256, query_position_rate=1.0, key_position_rate=1.29,
                            use_memory_mask=False, trainable_positional_encodings=False,
                            force_monotonic_attention=True, use_decoder_state_for_postnet_input=True,
                            max_positions=512, embedding_weight_std=0.1,
                            speaker_embedding_weight_std=0.01, freeze_embedding=False, window_ahead=3,
                            window_backward=1, key_projection=True, value_projection=True):
    """Build multi-speaker deepvoice3 (reference builder.py:172-258): one pre-attention layer,
    attention on the first decoder layer only."""
    return _dv3(n_vocab, (1,), [True, False, False, False, False], embed_dim, mel_dim, linear_dim, r,
                downsample_step, n_speakers, speaker_embed_dim, padding_idx, dropout, kernel_size,
                encoder_channels, decoder_channels, converter_channels, query_position_rate,
                key_position_rate, use_memory_mask, trainable_positional_encodings,
                force_monotonic_attention, use_decoder_state_for_postnet_input, max_positions,
                embedding_weight_std, speaker_embedding_weight_std, freeze_embedding, window_ahead,
                window_backward, key_projection, value_projection)


def nyanko(n_vocab, embed_dim=128, mel_dim=80, linear_dim=513, r=1, downsample_step=4, n_speakers=1,
           speaker_embed_dim=16, padding_idx=0, dropout=(1 - 0.95), kernel_size=3, encoder_channels=256,
           decoder_channels=256, converter_channels=512, query_position_rate=1.0, key_position_rate=1.29,
           use_memory_mask=False, trainable_positional_encodings=False, force_monotonic_attention=True,
           use_decoder_state_for_postnet_input=False, max_positions=512, embedding_weight_std=0.01,
           speaker_embedding_weight_std=0.01, freeze_embedding=False, window_ahead=3, window_backward=1,
           key_projection=False, value_projection=False):
    """Build nyanko (reference builder.py:96-169)."""
    from .nyanko import Encoder, Decoder, Converter
    assert encoder_channels == decoder_channels

    if n_speakers != 1:
        raise ValueError("Multi-speaker is not supported")
    if not (downsample_step == 4 and r == 1):
        raise ValueError("Not supported. You need to change hardcoded parameters")

    encoder = Encoder(n_vocab, embed_dim, channels=encoder_channels, kernel_size=kernel_size,
                      padding_idx=padding_idx, n_speakers=n_speakers, speaker_embed_dim=speaker_embed_dim,
                      dropout=dropout, embedding_weight_std=embedding_weight_std)
    decoder = Decoder(embed_dim, in_dim=mel_dim, r=r, channels=decoder_channels, kernel_size=kernel_size,
                      padding_idx=padding_idx, n_speakers=n_speakers, speaker_embed_dim=speaker_embed_dim,
                      dropout=dropout, max_positions=max_positions,
                      force_monotonic_attention=force_monotonic_attention,
                      query_position_rate=query_position_rate, key_position_rate=key_position_rate,
                      use_memory_mask=use_memory_mask, window_ahead=window_ahead,
                      window_backward=window_backward, key_projection=key_projection,
                      value_projection=value_projection)
    seq2seq = AttentionSeq2Seq(encoder, decoder)

    in_dim = decoder_channels // r if use_decoder_state_for_postnet_input else mel_dim
    converter = Converter(in_dim=in_dim, out_dim=linear_dim, channels=converter_channels,
                          kernel_size=kernel_size, dropout=dropout)

    return MultiSpeakerTTSModel(
        seq2seq, converter, padding_idx=padding_idx, mel_dim=mel_dim, linear_dim=linear_dim,
        n_speakers=n_speakers, speaker_embed_dim=speaker_embed_dim,
        trainable_positional_encodings=trainable_positional_encodings,
        use_decoder_state_for_postnet_input=use_decoder_state_for_postnet_input,
        speaker_embedding_weight_std=speaker_embedding_weight_std, freeze_embedding=freeze_embedding)
