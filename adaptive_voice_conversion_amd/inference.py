"""One-shot conversion front (reference: inference.py:24-93 ``Inferencer``), the immediate
caller of the hot path.

Kept from the reference: ``Inferencer(config, args)``, ``load_model`` (``args.model`` =
``<path>.ckpt`` state_dict), ``attr`` pickle with per-bin ``mean``/``std``,
``utt_make_frames``, ``normalize`` / ``denormalize``, ``inference_one_utterance(x, x_cond)``.
Added: ``convert_batch`` — many (source, target) pairs of arbitrary, unequal lengths in few engine
calls (pairs are bucketed by shape; the reference only ever runs batch 1).

The audio front and back end (reference: ``get_spectrograms`` / ``melspectrogram2wav`` of
``preprocess/tacotron/utils.py``, librosa on the CPU there) run on the GPU through ``dsp.MelDSP``:
``inference_from_path`` (inference.py:86-93) reads two wav files and writes the converted one, ``mel2wav`` defaults to the
device-side ``melspectrogram2wav`` when the model has as many mel bins as the DSP hyper-parameters produce (the stock
``config.yaml`` trains on 512-mel features, hyperparams.py:29); a caller-supplied ``mel2wav`` takes precedence, and with
neither the waveform slot of the result is ``None``.
"""
import pickle
from collections import defaultdict

import torch

from .model import AE
from .utils import cc, local_device


class Inferencer(object):
    def __init__(self, config, args, mel2wav=None, lib=None, dsp_hp=None):
        self.config = config
        self.args = args
        self.mel2wav = mel2wav
        self._lib = lib
        self._dsp_hp = dsp_hp   # None: preprocess/tacotron/hyperparams.py's values
        self.build_model()
        if getattr(args, "model", None):
            self.load_model()
        self.attr = None
        if getattr(args, "attr", None):
            with open(args.attr, "rb") as f:
                self.attr = pickle.load(f)
        self._dsp = None

    def dsp(self):
        """The device-side mel <-> waveform DSP (built on first use: its DFT bases take ~30 MB of HBM)."""
        if self._dsp is None:
            from .dsp import Hyperparams, MelDSP
            self._dsp = MelDSP(self._dsp_hp or Hyperparams, device=self.model.flat_parameters().device, lib=self._lib)
        return self._dsp

    def load_model(self):
        dev = self.model.flat_parameters().device
        self.model.load_state_dict(torch.load(f"{self.args.model}", map_location=dev))

    def build_model(self):
        self.model = AE(self.config, lib=self._lib) if self._lib is not None else cc(AE(self.config))
        self.model.eval()

    # ---- inference.py:54-60
    def utt_make_frames(self, x):
        frame_size = self.config["data_loader"]["frame_size"]
        remains = x.size(0) % frame_size
        if remains != 0:
            x = torch.nn.functional.pad(x, (0, remains))
        return x.view(1, x.size(0) // frame_size, frame_size * x.size(1)).transpose(1, 2)

    def denormalize(self, x):
        if self.attr is None:
            return x
        return x * self.attr["std"] + self.attr["mean"]

    def normalize(self, x):
        if self.attr is None:
            return x
        return (x - self.attr["mean"]) / self.attr["std"]

    # ---- inference.py:62-70
    def inference_one_utterance(self, x, x_cond):
        """x: [T, M] source mel, x_cond: [T', M] target-speaker mel (normalised).  Returns (wav | None, mel [T'', M])."""
        dev = self.model.flat_parameters().device
        x = self.utt_make_frames(x.to(dev))
        x_cond = self.utt_make_frames(x_cond.to(dev))
        with torch.no_grad():
            dec = self.model.inference(x, x_cond)
        dec = dec.transpose(1, 2).squeeze(0).detach().cpu().numpy()
        dec = self.denormalize(dec)
        if self.mel2wav is not None:
            wav = self.mel2wav(dec)
        elif getattr(self.args, "source", None) is not None and dec.shape[1] == self.dsp().hp.n_mels:
            wav = self.dsp().melspectrogram2wav(dec)            # inference.py:69
        else:
            wav = None
        return wav, dec

    # ---- inference.py:82-93
    def write_wav_to_file(self, wav_data, output_path):
        from scipy.io.wavfile import write
        write(output_path, rate=int(getattr(self.args, "sample_rate", 24000)), data=wav_data)

    def inference_from_path(self):
        dsp = self.dsp()
        if self.model._n_mels != dsp.hp.n_mels:
            raise ValueError(f"the model works on {self.model._n_mels}-mel features, get_spectrograms produces {dsp.hp.n_mels} "
                             "(preprocess/tacotron/hyperparams.py:29)")
        src_mel, _ = dsp.get_spectrograms(self.args.source)
        tar_mel, _ = dsp.get_spectrograms(self.args.target)
        src_mel = torch.from_numpy(self.normalize(src_mel)).float()
        tar_mel = torch.from_numpy(self.normalize(tar_mel)).float()
        conv_wav, conv_mel = self.inference_one_utterance(src_mel, tar_mel)
        self.write_wav_to_file(conv_wav, self.args.output)
        return conv_wav, conv_mel

    def convert_batch_to_wav(self, pairs, max_streams=4, do_trim=True, n_iter=None):
        """`convert_batch` + the audio back end: the converted mels are denormalised (inference.py:68) and vocoded by
        `melspectrogram2wav` -- all utterances, whatever their lengths, in ONE batched Griffin-Lim launch set (dsp.MelDSP).
        Returns (list of float32 waveforms, list of converted mels) in input order."""
        mels = [self.denormalize(m.numpy()) for m in self.convert_batch(pairs, max_streams)]
        return self.dsp().melspectrogram2wav_batch(mels, do_trim=do_trim, n_iter=n_iter), mels   # ONE Griffin-Lim launch set, any lengths

    def convert_batch(self, pairs, max_streams=4, ragged=True):
        """pairs: list of (src [T,M], tgt [T',M]) tensors of any lengths.  Default: ONE ragged launch set over all pairs
        (``AE.inference_ragged``: per-sample lengths inside every kernel; real utterances all differ in length, so shape buckets
        would be batches of one).  ``ragged=False``: the round-2 path -- pairs with equal (T, T') share one uniform plan,
        different shapes go out on up to ``max_streams`` HIP streams.  Lengths are never padded: reflect padding and the
        InstanceNorm statistics depend on the true length, so padding would change the result.
        Under ``compute_dtype: bf16`` the ragged path rounds the matrix-product operands to bf16 on fp32 storage ("bf16r"; the bf16
        pair-STORAGE engine of ``AE.inference`` takes uniform shapes only) -- ``self.model.last_ragged_compute`` says which mode ran.
        Returns the converted mels ([T'',M] CPU tensors) in input order."""
        if ragged:
            with torch.no_grad():
                outs = self.model.inference_ragged([s for s, _ in pairs], [t for _, t in pairs])
            return [o.t().cpu() for o in outs]
        return self._convert_batch_bucketed(pairs, max_streams)

    def _convert_batch_bucketed(self, pairs, max_streams=4):
        dev = self.model.flat_parameters().device
        buckets = defaultdict(list)
        for i, (s, t) in enumerate(pairs):
            buckets[(s.shape[0], t.shape[0])].append(i)
        out = [None] * len(pairs)
        cuda = dev.type == "cuda"
        streams = [torch.cuda.Stream(device=dev) for _ in range(min(max_streams, len(buckets)))] if cuda and len(buckets) > 1 else []
        main = torch.cuda.current_stream(dev) if cuda else None
        results = []
        with torch.no_grad():
            for k, ((_, _), idx) in enumerate(sorted(buckets.items(), key=lambda kv: -kv[0][0] * len(kv[1]))):   # big buckets first
                xs = torch.stack([pairs[i][0] for i in idx]).to(dev).transpose(1, 2)   # [B, M, T] views, no copy
                xc = torch.stack([pairs[i][1] for i in idx]).to(dev).transpose(1, 2)
                if streams:
                    st = streams[k % len(streams)]
                    st.wait_stream(main)                      # inputs were produced on the caller's stream
                    with torch.cuda.stream(st):
                        dec = self.model.inference(xs, xc)    # (one plan + workspace per shape: independent of other buckets)
                        xs.record_stream(st), xc.record_stream(st)
                        for e in self.model._plans.d["inference"].values():   # the workspace may have been allocated on another stream
                            if e.ws is not None:
                                e.ws.record_stream(st)
                else:
                    dec = self.model.inference(xs, xc)
                results.append((idx, dec))
            for st in streams:
                main.wait_stream(st)
            for idx, dec in results:
                dec = dec.transpose(1, 2).cpu()
                for k, i in enumerate(idx):
                    out[i] = dec[k]
        return out


def main(argv=None):
    """The reference's command line (inference.py:95-109)."""
    from argparse import ArgumentParser
    from .config import load_config
    parser = ArgumentParser()
    parser.add_argument('-attr', '-a', help='attr file path')
    parser.add_argument('-config', '-c', help='config file path')
    parser.add_argument('-model', '-m', help='model path')
    parser.add_argument('-source', '-s', help='source wav path')
    parser.add_argument('-target', '-t', help='target wav path')
    parser.add_argument('-output', '-o', help='output wav path')
    parser.add_argument('-sample_rate', '-sr', help='sample rate', default=24000, type=int)
    args = parser.parse_args(argv)
    inferencer = Inferencer(config=load_config(args.config), args=args)
    inferencer.inference_from_path()


if __name__ == '__main__':
    main()
