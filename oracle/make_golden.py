# coding: utf-8
"""Generate tests/golden/*.npz from the UNMODIFIED reference (build container only).

    python -m oracle.make_golden            # rewrites tests/golden/

TEST INFRASTRUCTURE ONLY.  Imports /root/reference through oracle/refimport.py, builds small
models with fixed seeds, runs the reference's own forward / incremental decode / loss / train
step on CPU, and stores inputs + state_dict + outputs.  tests/test_oracle_golden.py pins
oracle/dv3_oracle.py against these files; the -m gpu tests pin the HIP path against the oracle
and, through the same files, against the reference itself.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import refimport  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

# (name, builder, hyper-parameters).  Shapes are tiny so the fixtures stay small, but every
# structural feature of the presets is present: k=3 dilations 1..27, downsample_step 4 with two
# ConvTranspose1d, key/value projections, memory mask, multi-speaker softsign biases.
MODELS = [
    ("dv3_tiny", "deepvoice3", dict(
        n_vocab=40, embed_dim=32, mel_dim=80, linear_dim=65, r=4, padding_idx=0, dropout=0.05,
        kernel_size=5, encoder_channels=16, decoder_channels=32, converter_channels=32,
        force_monotonic_attention=False, use_decoder_state_for_postnet_input=False)),
    ("dv3_preset_like", "deepvoice3", dict(
        n_vocab=40, embed_dim=24, mel_dim=20, linear_dim=33, r=1, downsample_step=4, padding_idx=0,
        dropout=0.05, kernel_size=3, encoder_channels=48, decoder_channels=32,
        converter_channels=32, query_position_rate=1.0, key_position_rate=1.385,
        use_memory_mask=True, force_monotonic_attention=True,
        use_decoder_state_for_postnet_input=True, key_projection=True, value_projection=True)),
    ("dv3_multispeaker", "deepvoice3_multispeaker", dict(
        n_vocab=40, embed_dim=24, mel_dim=20, linear_dim=33, r=1, downsample_step=4, padding_idx=0,
        n_speakers=5, speaker_embed_dim=8, dropout=0.05, kernel_size=3, encoder_channels=48,
        decoder_channels=32, converter_channels=32, query_position_rate=2.0,
        key_position_rate=7.6, use_memory_mask=True, force_monotonic_attention=True,
        use_decoder_state_for_postnet_input=True, max_positions=128,
        speaker_embedding_weight_std=0.05)),
    ("nyanko_tiny", "nyanko", dict(
        n_vocab=40, embed_dim=16, mel_dim=20, linear_dim=33, r=1, downsample_step=4, padding_idx=0,
        dropout=0.05, kernel_size=3, encoder_channels=32, decoder_channels=32,
        converter_channels=32, use_memory_mask=True, force_monotonic_attention=True,
        use_decoder_state_for_postnet_input=True, max_positions=128)),
]


def _np(t):
    return t.detach().cpu().numpy()


def make_batch(rng, hp, B, Tt_max, Td, n_speakers=1):
    """Random padded batch in the conventions of train.collate_fn (train.py:293-360)."""
    r = hp.get("r", 4)
    ds = hp.get("downsample_step", 1)
    in_len = rng.randint(max(3, Tt_max // 2), Tt_max + 1, size=B)
    in_len[0] = Tt_max
    text = np.zeros((B, Tt_max), dtype=np.int64)
    text_pos = np.zeros((B, Tt_max), dtype=np.int64)
    for b in range(B):
        text[b, :in_len[b]] = rng.randint(2, hp["n_vocab"], size=in_len[b])
        text_pos[b, :in_len[b]] = np.arange(1, in_len[b] + 1)
    mel = rng.rand(B, Td * r, hp["mel_dim"]).astype(np.float32)
    frame_pos = np.tile(np.arange(1, Td + 1, dtype=np.int64)[None, :], (B, 1))
    spk = rng.randint(0, n_speakers, size=B).astype(np.int64) if n_speakers > 1 else None
    return dict(text=text, text_positions=text_pos, frame_positions=frame_pos, mel=mel,
                input_lengths=in_len.astype(np.int64), speaker_ids=spk)


def ref_forward(model, text, mel=None, speaker_ids=None, text_positions=None, frame_positions=None,
                input_lengths=None):
    """model(...) of the reference.  Under torch >= 2 the reference's own glue
    (`mel_outputs.view(B, -1, mel_dim)`, __init__.py:83) raises for r > 1 because sigmoid now
    preserves the transposed strides of its input; in that case the same glue is replayed here
    with .reshape (identical values -- it is what older torch computed)."""
    try:
        return model(text, mel, speaker_ids=speaker_ids, text_positions=text_positions,
                     frame_positions=frame_positions, input_lengths=input_lengths)
    except RuntimeError as e:
        if "view size is not compatible" not in str(e):
            raise
    B = text.size(0)
    se = model.embed_speakers(speaker_ids) if speaker_ids is not None else None
    mo, al, dn, st = model.seq2seq(text, mel, se, text_positions, frame_positions, input_lengths)
    mo = mo.reshape(B, -1, model.mel_dim)
    pin = st.reshape(B, mo.size(1), -1) if model.use_decoder_state_for_postnet_input else mo
    return mo, model.postnet(pin, se), al, dn


def gen_model(name, builder_name, hp):
    pkg = refimport.load_model_package()
    from deepvoice3_pytorch import builder
    torch.manual_seed(1234)
    model = getattr(builder, builder_name)(**hp)
    # make biases / speaker tables non-trivial so a swapped or dropped bias cannot pass
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".bias"):
                p.uniform_(-0.1, 0.1)
            if n.endswith("weight_g"):
                p.mul_(torch.empty_like(p).uniform_(0.8, 1.2))
    model.eval()
    rng = np.random.RandomState(7)
    nspk = hp.get("n_speakers", 1)
    B, Tt, Td = 3, 13, 9
    batch = make_batch(rng, hp, B, Tt, Td, nspk)
    out = {"hp": json.dumps(hp), "builder": builder_name}
    sd = {k: _np(v) for k, v in model.state_dict().items()}
    for k, v in sd.items():
        out["sd/" + k] = v
    for k, v in batch.items():
        if v is not None:
            out["in/" + k] = v
    text = torch.from_numpy(batch["text"])
    mel = torch.from_numpy(batch["mel"])
    tp = torch.from_numpy(batch["text_positions"])
    fp = torch.from_numpy(batch["frame_positions"])
    spk = torch.from_numpy(batch["speaker_ids"]) if nspk > 1 else None
    with torch.no_grad():
        mo, lo, al, dn = ref_forward(model, text, mel, speaker_ids=spk, text_positions=tp,
                                     frame_positions=fp, input_lengths=batch["input_lengths"])
    out.update({"out/mel": _np(mo), "out/linear": _np(lo), "out/alignments": _np(al),
                "out/done": _np(dn)})
    # encoder outputs (intermediate pin)
    with torch.no_grad():
        se = model.embed_speakers(spk) if nspk > 1 else None
        keys, values = model.seq2seq.encoder(text, lengths=batch["input_lengths"], speaker_embed=se)
    out.update({"out/enc_keys": _np(keys), "out/enc_values": _np(values)})
    # incremental decode, teacher forced (tests/test_deepvoice3.py:184-235 logic)
    dec = model.seq2seq.decoder
    r = hp.get("r", 4)
    if nspk > 1:
        # the reference's multi-speaker incremental path only works for B == 1: with B > 1
        # `a + softsign(speaker_proj(speaker_embed))` broadcasts (B,1,C)+(B,C) -> (B,B,C)
        # (modules.py:158-162 with the 2-D speaker_embed of deepvoice3.py:417).  Use item 0.
        B = 1
        text, mel, tp, fp, spk = text[:1], mel[:1], tp[:1], fp[:1], spk[:1]
        keys, values, se = keys[:1], values[:1], se[:1]
    out["inc_batch"] = np.int64(B)
    mel_r = mel.view(B, mel.size(1) // r, -1)
    with torch.no_grad():
        dec.start_fresh_sequence()
        if builder_name == "nyanko":
            io, ia, idn, ist = dec.incremental_forward((keys, values), tp, test_inputs=mel_r)
        else:
            io, ia, idn, ist = dec.incremental_forward((keys, values), tp, speaker_embed=se,
                                                       test_inputs=mel_r)
    out.update({"inc_tf/mel": _np(io), "inc_tf/alignments": _np(ia), "inc_tf/states": _np(ist),
                "inc_tf/done": _np(torch.stack(idn))})
    # free-running generation with a fixed number of steps (issue38 path)
    dec.max_decoder_steps = 12
    dec.min_decoder_steps = 12
    with torch.no_grad():
        g_mo, g_lo, g_al, g_dn = ref_forward(model, text, speaker_ids=spk, text_positions=tp)
        g_mo2, _, _, _ = ref_forward(model, text, speaker_ids=spk, text_positions=tp)
    assert (g_mo == g_mo2).all()
    out.update({"gen/mel": _np(g_mo), "gen/linear": _np(g_lo), "gen/alignments": _np(g_al),
                "gen/done": _np(torch.stack(g_dn))})
    np.savez_compressed(os.path.join(OUT, "model_%s.npz" % name), **out)
    print("wrote model_%s.npz: %d tensors, mel %s linear %s" % (name, len(sd), tuple(mo.shape), tuple(lo.shape)))


def gen_losses():
    train, hparams, _ = refimport.load_train_module()
    rng = np.random.RandomState(3)
    B, T, D, r = 3, 11, 7, 1
    y_hat = torch.from_numpy(rng.uniform(0.02, 0.98, size=(B, T, D)).astype(np.float32))
    y = torch.from_numpy(rng.rand(B, T, D).astype(np.float32))
    lengths = torch.tensor([11, 7, 4]).long()
    mask = train.sequence_mask(lengths, max_len=T).unsqueeze(-1)[:, r:, :]
    out = {}
    for wm, wbd in [(0.5, 0.1), (0.0, 0.1), (0.5, 0.0)]:
        hparams.masked_loss_weight = wm
        hparams.binary_divergence_weight = wbd
        yh = y_hat.clone().requires_grad_(True)
        l1, bd = train.spec_loss(yh[:, :-r, :], y[:, r:, :], mask if wm > 0 else None)
        total = (1 - wbd) * l1 + wbd * bd
        total.sum().backward()
        tag = "spec_wm%g_wbd%g" % (wm, wbd)
        out[tag + "/l1"] = _np(l1)
        out[tag + "/bd"] = _np(bd.reshape(-1)[0])
        out[tag + "/grad"] = _np(yh.grad)
    out.update({"spec/y_hat": _np(y_hat), "spec/y": _np(y), "spec/lengths": _np(lengths)})
    # priority-frequency L1 (train.py:562-569): first 3 of the 7 bins weighted 0.3
    hparams.masked_loss_weight, hparams.binary_divergence_weight = 0.5, 0.1
    yh = y_hat.clone().requires_grad_(True)
    l1, bd = train.spec_loss(yh[:, :-r, :], y[:, r:, :], mask, priority_bin=3, priority_w=0.3)
    ((1 - 0.1) * l1 + 0.1 * bd).sum().backward()
    out.update({"spec_priority/l1": _np(l1), "spec_priority/bd": _np(bd.reshape(-1)[0]), "spec_priority/grad": _np(yh.grad)})
    hparams.masked_loss_weight = 0.5
    hparams.binary_divergence_weight = 0.1
    il, tl = np.array([9, 5, 7]), np.array([12, 8, 3])
    for g in (0.2, 0.4):
        out["guided_g%g" % g] = train.guided_attentions(il, tl, 12, g=g)
    out["guided/in_len"], out["guided/out_len"] = il, tl
    p = torch.from_numpy(rng.uniform(0.01, 0.99, size=(4, 6, 1)).astype(np.float32)).requires_grad_(True)
    t = torch.from_numpy((rng.rand(4, 6, 1) > 0.5).astype(np.float32))
    l = torch.nn.BCELoss()(p, t)
    l.backward()
    out.update({"bce/p": _np(p), "bce/t": _np(t), "bce/loss": _np(l), "bce/grad": _np(p.grad)})
    for step in (0, 10, 3999, 4000, 100000):
        out["noam/%d" % step] = np.float64(__import__("lrschedule").noam_learning_rate_decay(5e-4, step))
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)
    print("wrote losses.npz")


def gen_trainstep():
    """Two steps of the reference's own train.train() on CPU with dropout=0 (deterministic)."""
    train, hparams, Writer = refimport.load_train_module()
    import deepvoice3_pytorch.frontend as fe
    hparams.parse_json(open(os.path.join(refimport.REF_ROOT, "presets", "deepvoice3_ljspeech.json")).read())
    hp_over = dict(dropout=0.0, text_embed_dim=24, encoder_channels=48, decoder_channels=32,
                   converter_channels=32, num_mels=20, fft_size=64, batch_size=3, max_positions=128)
    for k, v in hp_over.items():
        setattr(hparams, k, v)
    train._frontend = fe.en
    torch.manual_seed(4321)
    model = train.build_model()
    sd0 = {k: _np(v).copy() for k, v in model.state_dict().items()}
    rng = np.random.RandomState(11)
    items = []
    for n_frames, n_text in [(37, 9), (52, 13), (44, 11)]:
        text = np.concatenate([rng.randint(2, 149, size=n_text - 1), [1]]).astype(np.int32)
        items.append((text, rng.rand(n_frames, 20).astype(np.float32),
                      rng.rand(n_frames, 33).astype(np.float32)))
    batch = train.collate_fn(items)
    optimizer = torch.optim.Adam(model.get_trainable_parameters(), lr=hparams.initial_learning_rate,
                                 betas=(hparams.adam_beta1, hparams.adam_beta2), eps=hparams.adam_eps,
                                 weight_decay=hparams.weight_decay, amsgrad=hparams.amsgrad)
    writer = Writer()
    # start at the top of the Noam warm-up so the two Adam updates are well above fp32 noise
    train.global_step, train.global_epoch = 3999, 0
    train.train(torch.device("cpu"), model, [batch, batch], optimizer, writer,
                init_lr=hparams.initial_learning_rate, checkpoint_dir="/tmp",
                checkpoint_interval=10 ** 9, nepochs=1, clip_thresh=hparams.clip_thresh)
    out = {"hp_over": json.dumps(hp_over), "global_step0": np.int64(3999)}
    for k, v in sd0.items():
        out["sd0/" + k] = v
    for k, v in model.state_dict().items():
        out["sd2/" + k] = _np(v)
    x, in_len, mel, y, (tp, fp), done, tgt_len, _ = batch
    out.update({"in/text": _np(x), "in/input_lengths": _np(in_len), "in/mel": _np(mel), "in/y": _np(y),
                "in/text_positions": _np(tp), "in/frame_positions": _np(fp), "in/done": _np(done),
                "in/target_lengths": _np(tgt_len)})
    for k, v in writer.scalars.items():
        out["scalar/" + k.replace(" ", "_")] = np.array([s[1] for s in v], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "trainstep.npz"), **out)
    print("wrote trainstep.npz; scalars:", {k: [round(s[1], 6) for s in v] for k, v in writer.scalars.items()})


def gen_trainstep_split():
    """Two steps each of the reference's own train.train(train_seq2seq=True, train_postnet=False) and of the converse
    (train.py:608-616, 684-731, 788-797) on the trainstep fixture's model and batch (dropout 0), plus the checkpoint
    files train.save_checkpoint writes for the two modes ("_seq2seq" / "_postnet": the sub-module's state_dict and the
    full optimizer's state, which holds moments only for the parameters that received gradients)."""
    train, hparams, Writer = refimport.load_train_module()
    import deepvoice3_pytorch.frontend as fe
    hparams.parse_json(open(os.path.join(refimport.REF_ROOT, "presets", "deepvoice3_ljspeech.json")).read())
    hp_over = dict(dropout=0.0, text_embed_dim=24, encoder_channels=48, decoder_channels=32,
                   converter_channels=32, num_mels=20, fft_size=64, batch_size=3, max_positions=128)
    for k, v in hp_over.items():
        setattr(hparams, k, v)
    # use_decoder_state_for_postnet_input is True in the preset: the post-net then has decoder_channels // r inputs and
    # can not be trained on mel targets alone (model.postnet(mel), train.py:701) -- the split modes need the mel-input form
    hparams.use_decoder_state_for_postnet_input = False
    train._frontend = fe.en
    rng = np.random.RandomState(11)
    items = []
    for n_frames, n_text in [(37, 9), (52, 13), (44, 11)]:
        text = np.concatenate([rng.randint(2, 149, size=n_text - 1), [1]]).astype(np.int32)
        items.append((text, rng.rand(n_frames, 20).astype(np.float32), rng.rand(n_frames, 33).astype(np.float32)))
    batch = train.collate_fn(items)
    out = {"hp_over": json.dumps(dict(hp_over, use_decoder_state_for_postnet_input=False)), "global_step0": np.int64(3999)}
    x, in_len, mel, y, (tp, fp), done, tgt_len, _ = batch
    out.update({"in/text": _np(x), "in/input_lengths": _np(in_len), "in/mel": _np(mel), "in/y": _np(y),
                "in/text_positions": _np(tp), "in/frame_positions": _np(fp), "in/done": _np(done),
                "in/target_lengths": _np(tgt_len)})
    import tempfile
    for mode, (ts, tpn) in (("seq2seq", (True, False)), ("postnet", (False, True))):
        torch.manual_seed(4321)
        model = train.build_model()
        if mode == "seq2seq":
            for k, v in model.state_dict().items():
                out["sd0/" + k] = _np(v).copy()
        optimizer = torch.optim.Adam(model.get_trainable_parameters(), lr=hparams.initial_learning_rate,
                                     betas=(hparams.adam_beta1, hparams.adam_beta2), eps=hparams.adam_eps,
                                     weight_decay=hparams.weight_decay, amsgrad=hparams.amsgrad)
        writer = Writer()
        train.global_step, train.global_epoch = 3999, 0
        train.train(torch.device("cpu"), model, [batch, batch], optimizer, writer,
                    init_lr=hparams.initial_learning_rate, checkpoint_dir="/tmp", checkpoint_interval=10 ** 9, nepochs=1,
                    clip_thresh=hparams.clip_thresh, train_seq2seq=ts, train_postnet=tpn)
        for k, v in model.state_dict().items():
            out["%s/sd2/%s" % (mode, k)] = _np(v)
        for k, v in writer.scalars.items():
            out["%s/scalar/%s" % (mode, k.replace(" ", "_"))] = np.array([s[1] for s in v], dtype=np.float64)
        # the checkpoint the reference writes in this mode: names of its state_dict, which optimizer slots carry state
        d = tempfile.mkdtemp()
        hparams.save_optimizer_state = True
        train.save_checkpoint(model, optimizer, train.global_step, d, train.global_epoch, ts, tpn)
        fn = os.listdir(d)[0]
        ck = torch.load(os.path.join(d, fn), map_location="cpu", weights_only=False)
        out["%s/ckpt_name" % mode] = fn
        out["%s/ckpt_keys" % mode] = json.dumps(sorted(ck["state_dict"].keys()))
        out["%s/ckpt_opt_slots" % mode] = np.array(sorted(ck["optimizer"]["state"].keys()), dtype=np.int64)
        out["%s/ckpt_opt_nparams" % mode] = np.int64(len(ck["optimizer"]["param_groups"][0]["params"]))
        print(mode, fn, "scalars:", {k: [round(s[1], 6) for s in v] for k, v in writer.scalars.items()})
    np.savez_compressed(os.path.join(OUT, "trainstep_split.npz"), **out)
    print("wrote trainstep_split.npz")


def gen_misc():
    pkg = refimport.load_model_package()
    from deepvoice3_pytorch.modules import position_encoding_init, SinusoidalEncoding
    out = {"pe/table_r1.0": _np(position_encoding_init(64, 24, position_rate=1.0)),
           "pe/table_r1.385": _np(position_encoding_init(64, 24, position_rate=1.385)),
           "pe/raw": _np(position_encoding_init(64, 24, position_rate=1.0, sinusoidal=False))}
    se = SinusoidalEncoding(64, 24)
    pos = torch.tensor([[1, 2, 3, 0], [5, 60, 7, 8]]).long()
    out["pe/pos"] = _np(pos)
    out["pe/enc_w1.385"] = _np(se(pos, 1.385))
    out["pe/enc_wvec"] = _np(se(pos, torch.tensor([0.7, 2.3])))
    np.savez_compressed(os.path.join(OUT, "misc.npz"), **out)
    print("wrote misc.npz")


def gen_collate():
    """The reference's own train.collate_fn on ragged items, single- and multi-speaker, r in {1, 2}."""
    train, hparams, _ = refimport.load_train_module()
    out = {}
    rng = np.random.RandomState(21)
    for case, (r, ds, spk) in enumerate([(1, 4, False), (2, 4, True), (1, 1, False), (4, 1, True)]):
        hparams.outputs_per_step, hparams.downsample_step = r, ds
        items = []
        for n_frames, n_text in [(37, 9), (52, 13), (8, 2), (44, 11)]:
            it = (np.concatenate([rng.randint(2, 149, size=n_text - 1), [1]]).astype(np.int64),
                  rng.rand(n_frames, 5).astype(np.float32), rng.rand(n_frames, 7).astype(np.float32))
            items.append(it + ((int(rng.randint(0, 10)),) if spk else ()))
        x, in_len, mel, y, (tp, fp), done, tgt_len, sid = train.collate_fn(items)
        pre = "c%d/" % case
        out[pre + "r_ds"] = np.array([r, ds], dtype=np.int64)
        for i, it in enumerate(items):
            out[pre + "item%d/text" % i], out[pre + "item%d/mel" % i], out[pre + "item%d/y" % i] = it[0], it[1], it[2]
            if spk:
                out[pre + "item%d/spk" % i] = np.int64(it[3])
        for k, v in dict(x=x, input_lengths=in_len, mel=mel, y=y, text_positions=tp, frame_positions=fp,
                         done=done, target_lengths=tgt_len).items():
            out[pre + "out/" + k] = _np(v)
        if spk:
            out[pre + "out/speaker_ids"] = _np(sid)
    np.savez_compressed(os.path.join(OUT, "collate.npz"), **out)
    print("wrote collate.npz")


def gen_audio_helpers():
    """The reference's own amplitude / dB / normalisation helpers (audio.py:75-93), run unmodified with the
    ljspeech preset's hparams: the part of the audio path that does not need the third-party lws / librosa /
    nnmnkwii packages (those are stubbed by refimport, so spectrogram / inv_spectrogram themselves cannot run)."""
    train, hparams, _ = refimport.load_train_module()
    hparams.parse_json(open(os.path.join(refimport.REF_ROOT, "presets", "deepvoice3_ljspeech.json")).read())
    import audio                                   # the reference audio.py, unmodified
    rng = np.random.RandomState(11)
    amp = np.concatenate([np.abs(rng.randn(64)) * 3.0, [0.0, 1e-7, 1e-5, 1.0, 37.5]]).astype(np.float64)
    amp32 = amp.astype(np.float32)
    db = np.concatenate([rng.uniform(-130, 30, 64), [-100.0, 0.0, -99.999, 20.0]]).astype(np.float64)
    norm = np.concatenate([rng.uniform(-0.3, 1.3, 64), [0.0, 1.0, 0.5]]).astype(np.float32)
    out = {"hp/min_level_db": np.float64(hparams.min_level_db), "hp/ref_level_db": np.float64(hparams.ref_level_db),
           "hp/power": np.float64(hparams.power), "hp/preemphasis": np.float64(hparams.preemphasis),
           "in/amp": amp, "in/amp32": amp32, "in/db": db, "in/norm": norm,
           "out/amp_to_db": audio._amp_to_db(amp), "out/amp_to_db32": audio._amp_to_db(amp32),
           "out/db_to_amp": audio._db_to_amp(db), "out/normalize": audio._normalize(db),
           "out/denormalize": audio._denormalize(norm),
           # the magnitude chain of inv_spectrogram up to the lws call (audio.py:39-41)
           "out/inv_mag": audio._db_to_amp(audio._denormalize(norm) + hparams.ref_level_db).astype(np.float64) ** hparams.power,
           # and of spectrogram after the STFT (audio.py:33-34)
           "out/spec_norm": audio._normalize(audio._amp_to_db(amp) - hparams.ref_level_db)}
    np.savez_compressed(os.path.join(OUT, "audio_helpers.npz"), **out)
    print("audio_helpers", len(out))


def gen_checkpoint():
    """A checkpoint file written by the reference's OWN train.save_checkpoint (train.py:788-809) after one
    train.train() step (weights + torch.optim.Adam state + counters), and the weights the reference reaches one
    step later from it.  tests: the file loads into train_step.Trainer and the next step lands on those weights."""
    import glob
    import shutil
    import tempfile
    train, hparams, Writer = refimport.load_train_module()
    import deepvoice3_pytorch.frontend as fe
    hparams.parse_json(open(os.path.join(refimport.REF_ROOT, "presets", "deepvoice3_ljspeech.json")).read())
    hp_over = dict(dropout=0.0, text_embed_dim=24, encoder_channels=48, decoder_channels=32,
                   converter_channels=32, num_mels=20, fft_size=64, batch_size=3, max_positions=128)
    for k, v in hp_over.items():
        setattr(hparams, k, v)
    train._frontend = fe.en
    torch.manual_seed(777)
    model = train.build_model()
    rng = np.random.RandomState(13)
    items = []
    for n_frames, n_text in [(41, 10), (33, 8), (52, 12)]:
        text = np.concatenate([rng.randint(2, 149, size=n_text - 1), [1]]).astype(np.int32)
        items.append((text, rng.rand(n_frames, 20).astype(np.float32), rng.rand(n_frames, 33).astype(np.float32)))
    batch = train.collate_fn(items)
    optimizer = torch.optim.Adam(model.get_trainable_parameters(), lr=hparams.initial_learning_rate,
                                 betas=(hparams.adam_beta1, hparams.adam_beta2), eps=hparams.adam_eps,
                                 weight_decay=hparams.weight_decay, amsgrad=hparams.amsgrad)
    tmp = tempfile.mkdtemp()
    train.global_step, train.global_epoch = 3999, 0
    kw = dict(init_lr=hparams.initial_learning_rate, checkpoint_dir=tmp, checkpoint_interval=10 ** 9,
              clip_thresh=hparams.clip_thresh)
    train.train(torch.device("cpu"), model, [batch], optimizer, Writer(), nepochs=1, **kw)
    train.save_checkpoint(model, optimizer, train.global_step, tmp, train.global_epoch, True, True)
    (path,) = glob.glob(os.path.join(tmp, "checkpoint_step*.pth"))
    dst = os.path.join(OUT, "reference_" + os.path.basename(path))
    shutil.copy(path, dst)
    writer = Writer()
    train.train(torch.device("cpu"), model, [batch], optimizer, writer, nepochs=2, **kw)    # one more step
    out = {"hp_over": json.dumps(hp_over), "ckpt_file": os.path.basename(dst), "global_step": np.int64(train.global_step)}
    for k, v in model.state_dict().items():
        out["sd_next/" + k] = _np(v)
    x, in_len, mel, y, (tp, fp), done, tgt_len, _ = batch
    out.update({"in/text": _np(x), "in/input_lengths": _np(in_len), "in/mel": _np(mel), "in/y": _np(y),
                "in/text_positions": _np(tp), "in/frame_positions": _np(fp), "in/done": _np(done),
                "in/target_lengths": _np(tgt_len)})
    for k, v in writer.scalars.items():
        out["scalar/" + k.replace(" ", "_")] = np.array([s[1] for s in v], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "checkpoint_resume.npz"), **out)
    shutil.rmtree(tmp)
    print("wrote", dst, "and checkpoint_resume.npz; step after:", train.global_step)


def gen_presets():
    """Preset-size outputs of the unmodified reference (BASELINE.json configs 2-4 at their real channel
    counts, bench-shaped batch B=2, Tt=150, 800 frames).  The 100 MB state_dicts are not stored: the
    weights are regenerated on both sides from (seed, per-tensor mean/std) by tests/util.synth_state_dict;
    the statistics are those of the reference's own initialisation."""
    import bench
    from tests.util import synth_state_dict
    from tests.test_gpu_preset_scale import _batch
    refimport.load_model_package()
    from deepvoice3_pytorch import builder
    for preset, (bname, hp, _) in sorted(bench.PRESETS.items()):
        hp = dict(hp)
        torch.manual_seed(99)
        model = getattr(builder, bname)(**hp)
        sd0 = model.state_dict()
        stats = {}
        for k, v in sd0.items():
            if k.endswith("positions.weight"):
                continue                      # frozen tables: deterministic, kept as built
            v = v.double()
            std = float(v.std()) if v.numel() > 1 else 0.0
            if k.endswith(".bias") and std == 0.0:
                std = 0.05                    # the reference zero-initialises biases: make them matter
            stats[k] = [float(v.mean()), std]
        seed, batch_seed = 20260922, 5
        sd = synth_state_dict({k: tuple(v.shape) for k, v in sd0.items()}, stats, seed, keep=sd0)
        model.load_state_dict(sd)
        model.eval()
        bt, spk = _batch(hp, seed=batch_seed)
        mel_ds = bt["mel"][:, 0::4, :].contiguous()
        with torch.no_grad():
            mo, lo, al, dn = ref_forward(model, bt["text"], mel_ds, speaker_ids=spk,
                                         text_positions=bt["text_positions"],
                                         frame_positions=bt["frame_positions"],
                                         input_lengths=bt["input_lengths"])
        out = {"sd_stats": json.dumps(stats), "seed": np.int64(seed), "batch_seed": np.int64(batch_seed),
               "out/mel": _np(mo), "out/linear_8": _np(lo[:, ::8].contiguous()), "out/alignments": _np(al),
               "out/done": _np(dn)}
        np.savez_compressed(os.path.join(OUT, "preset_%s.npz" % preset), **out)
        print("wrote preset_%s.npz: mel %s linear %s align %s" % (preset, tuple(mo.shape), tuple(lo.shape),
                                                                   tuple(al.shape)))


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "presets":
        gen_presets()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "checkpoint":
        gen_checkpoint()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "trainstep_split":
        gen_trainstep_split()
        return
    for name, b, hp in MODELS:
        gen_model(name, b, hp)
    gen_losses()
    gen_misc()
    gen_trainstep()
    gen_trainstep_split()
    gen_collate()
    gen_audio_helpers()
    gen_presets()
    gen_checkpoint()


if __name__ == "__main__":
    main()
