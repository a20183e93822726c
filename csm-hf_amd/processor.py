"""`CSMProcessor` -- conversation -> `[B, S, 33]` tensors, the producer of the generation path's inputs
(SURVEY.md section 8 f-1).  Same call signature and tensor layout as the reference
(`/root/reference/processor.py:41-378`); host-side Python, no GPU work.

Frame layout (reference `processor.py:200-378`):
  * text of a message -> `tokenizer.encode(f"[{speaker}]{text}", add_special_tokens=True)`; one frame per token,
    id in column 32, mask only on column 32 (`:254-267`);
  * audio of a message -> `audio_tokenizer.encode(wav[None, None])[0]` = `[32, F]` codes, plus ONE all-zero EOS
    frame; F+1 frames with mask on columns 0..31 (`:284-298`);
  * truncation keeps the LAST `max_length` frames (`:318-320`); batches are LEFT-padded (`:137-169`);
  * labels: -100 where masked and on the text column, message-level masking, decoder-label amortisation
    (1/ratio random frames keep codebooks 1..31; `random.sample`, so seed `random` for reproducibility).

Deviations (documented):
  * `text=..., speaker_id=...` works (the reference forwards its arguments positionally into the wrong
    slots and raises "Unsupported return format: True");
  * the padded-batch path returns an int32 `attention_mask` like the single-conversation path (the reference
    returns float32 there, which crashes a bf16 model through dtype promotion, `modeling_csm.py:328-332`).
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional, Union

import torch


class CSMProcessor:
    def __init__(self, tokenizer, audio_tokenizer):
        self.tokenizer = tokenizer
        self.audio_tokenizer = audio_tokenizer
        self.sample_rate = getattr(audio_tokenizer, "sample_rate", 16000)   # reference :52 (Mimi is 24 kHz)

    def __call__(self, messages=None, text=None, audios=None, speaker_id=None, return_tensors="pt", padding: bool = True,
                 truncation: bool = True, max_length: int = 2048, amortize_decoder_training: bool = True,
                 amortization_ratio: int = 16,
                 messages_training_mask: Optional[Union[List[int], List[bool], List[List[int]], List[List[bool]]]] = None
                 ) -> Dict[str, torch.Tensor]:
        if return_tensors != "pt":
            raise ValueError(f"Unsupported return format: {return_tensors}")
        if messages is None:
            if text is None or speaker_id is None:
                raise ValueError("Must provide either 'messages' or both 'text' and 'speaker_id'.")
            messages = [{"role": f"speaker_{speaker_id}", "content": [{"type": "text", "text": text}]}]
        is_batched = isinstance(messages[0], list) if messages else False
        if not is_batched:
            messages = [messages]
            audios = [audios] if audios is not None else [None]
            if messages_training_mask is not None:
                if isinstance(messages_training_mask[0], list):
                    raise ValueError("`messages_training_mask` is nested but expected flat for a single conversation.")
                messages_training_mask = [messages_training_mask]
        elif audios is not None and not isinstance(audios[0], list):
            audios = [audios]
        if audios is None:
            audios = []
        outs = []
        for i, convo in enumerate(messages):
            convo_mask = None
            if messages_training_mask is not None:
                if i >= len(messages_training_mask):
                    raise ValueError(f"messages_training_mask has {len(messages_training_mask)} entries but "
                                     f"{len(messages)} conversations were provided.")
                convo_mask = messages_training_mask[i]
            outs.append(self._process_messages(convo, audios[i] if i < len(audios) else None, truncation, max_length,
                                               amortize_decoder_training, amortization_ratio, convo_mask))
        if not outs:
            return {"input_ids": torch.zeros(0, 0, 33, dtype=torch.long), "attention_mask": torch.zeros(0, 0, 33, dtype=torch.int),
                    "labels": torch.zeros(0, 0, 33, dtype=torch.long)}
        S = max(o["input_ids"].size(0) for o in outs)
        ids, masks, labels = [], [], []
        for o in outs:
            n = o["input_ids"].size(0)
            if n < S and padding:      # left padding, reference :146-156
                pi = torch.zeros(S, 33, dtype=torch.long)
                pm = torch.zeros(S, 33, dtype=torch.int)
                pl = torch.full((S, 33), -100, dtype=torch.long)
                pi[S - n:], pm[S - n:], pl[S - n:] = o["input_ids"], o["attention_mask"], o["labels"]
                o = {"input_ids": pi, "attention_mask": pm, "labels": pl}
            ids.append(o["input_ids"].unsqueeze(0))
            masks.append(o["attention_mask"].unsqueeze(0))
            labels.append(o["labels"].unsqueeze(0))
        return {"input_ids": torch.cat(ids, 0), "attention_mask": torch.cat(masks, 0), "labels": torch.cat(labels, 0)}

    def _audio_device(self):
        try:
            return next(self.audio_tokenizer.parameters()).device
        except (StopIteration, AttributeError, TypeError):
            return torch.device("cpu")

    def _process_messages(self, messages, audios, truncation, max_length, amortize, ratio, training_mask):
        device = self._audio_device()
        toks, masks, bounds = [], [], []
        audio_index = 0
        n_frames = 0
        for mi, message in enumerate(messages):
            speaker = int(message["role"].split("_")[-1])
            keep = True if training_mask is None else bool(training_mask[mi])
            texts, has_audio = [], False
            for item in message["content"]:
                if item["type"] == "text" and item.get("text", ""):
                    texts.append(item["text"])
                elif item["type"] == "audio":
                    has_audio = True
            text = " ".join(texts)
            start = n_frames
            if text:
                tt = self.tokenizer.encode(f"[{speaker}]{text}", add_special_tokens=True)
                fr = torch.zeros(len(tt), 33, dtype=torch.long)
                fm = torch.zeros(len(tt), 33, dtype=torch.int)
                fr[:, -1] = torch.tensor(tt, dtype=torch.long)
                fm[:, -1] = 1
                toks.append(fr)
                masks.append(fm)
                n_frames += len(tt)
            if has_audio and audios and audio_index < len(audios) and audios[audio_index] is not None:
                wav = audios[audio_index]
                audio_index += 1
                if not isinstance(wav, torch.Tensor):
                    raise ValueError(f"Audio must be torch.Tensor, got {type(wav)}")
                with torch.no_grad():
                    codes = self.audio_tokenizer.encode(wav.unsqueeze(0).unsqueeze(0).to(device))[0]   # [32, F]
                codes = torch.cat([codes, torch.zeros(codes.size(0), 1, device=codes.device, dtype=codes.dtype)], dim=1)
                F = codes.size(1)
                fr = torch.zeros(F, 33, dtype=torch.long)
                fm = torch.zeros(F, 33, dtype=torch.int)
                fr[:, :-1] = codes.transpose(0, 1).to("cpu", torch.long)
                fm[:, :-1] = 1
                toks.append(fr)
                masks.append(fm)
                n_frames += F
            elif has_audio:
                print(f"Warning: Audio content declared but no audio tensor provided for message with "
                      f"{message.get('role', 'unknown')}")
            bounds.append((start, n_frames, keep))
        if audios and audio_index < len(audios):
            print(f"Warning: {len(audios) - audio_index} audio tensors were not used")
        if toks:
            tokens, tmask = torch.cat(toks, 0), torch.cat(masks, 0)
            if truncation and tokens.size(0) > max_length:
                tokens, tmask = tokens[-max_length:], tmask[-max_length:]
        else:
            tokens, tmask = torch.zeros(0, 33, dtype=torch.long), torch.zeros(0, 33, dtype=torch.int)
        labels = tokens.clone().masked_fill(tmask == 0, -100)
        labels[:, -1] = -100
        # message-level masking uses PRE-truncation indices, exactly like the reference (:335-341)
        for s, e, keep in bounds:
            if s >= labels.size(0):
                break
            e = min(e, labels.size(0))
            if not keep:
                labels[s:e, :] = -100
        if amortize:
            S = labels.shape[0]
            valid = torch.where(torch.any(labels[:, :-1] != -100, dim=-1))[0]
            frame_mask = torch.zeros(S, dtype=torch.bool)
            if len(valid) > 0:
                frame_mask[random.sample(valid.tolist(), max(1, len(valid) // ratio))] = True
            keep_mask = torch.zeros_like(labels, dtype=torch.bool)
            keep_mask[:, -1] = True
            keep_mask[:, 0:1] = torch.any(labels != -100, dim=-1, keepdim=True)
            keep_mask[frame_mask, 1:-1] = True
            labels = torch.where((labels != -100) & ~keep_mask, torch.full_like(labels, -100), labels)
        return {"input_ids": tokens, "attention_mask": tmask, "labels": labels}
