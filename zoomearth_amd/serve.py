"""OpenAI-compatible chat-completions shim over the engine, so the reference's `src/eval/infer_vllm.py` client
(`OpenAI(base_url="http://localhost:8000/v1").chat.completions.create(model=..., messages=[...])`,
/root/reference/src/eval/infer_vllm.py:19-24,153-156,218-221) can target this engine instead of `vllm serve`
(SURVEY.md 8f rank 3).

Request: the subset that client sends -- `messages` with `role` in {system, user, assistant} and `content` either a
string or a list of `{"type": "text", "text": ...}` / `{"type": "image_url", "image_url": {"url": "data:image/...;
base64,..."}}` items, plus `max_tokens`, `temperature`, `seed` (`n` must be 1, `stream` is not offered).
Prompt: the Qwen2.5-VL chat template (`<|im_start|>role\\n ... <|im_end|>\\n`, an image item becomes
`<|vision_start|><|image_pad|><|vision_end|>`, a default system turn when the conversation has none, then the
generation prompt `<|im_start|>assistant\\n`).  temperature 0 / absent -> greedy, else temperature sampling.
Concurrent requests (infer_vllm.py keeps up to 100 in flight, :244-271) share the GPU through continuous batching
(`zoomearth_amd/scheduler.py`): a dispatcher thread admits greedy requests into the RUNNING batch between bursts of
decode steps -- newcomers are prefilled together and join the next burst, a finished request frees its KV slot and its
response returns at once -- with each request's tokens independent of which other requests share its steps; sampled
requests run alone (temperature and seed are part of the captured decode step).  Images are decoded on the host and
uploaded.

    python -m zoomearth_amd.serve --model_name /ckpt/ZoomEarth-3B --port 8000
"""

import base64
import io
import threading
import time
import uuid

DEFAULT_SYSTEM = "You are a helpful assistant."
IMAGE_PLACEHOLDER = "<|vision_start|><|image_pad|><|vision_end|>"


class BadRequest(ValueError):
    pass


def decode_data_url(url: str):
    """data:image/...;base64,<payload> -> PIL RGB image (what infer_vllm.py's encode_pil_image_to_data_url produces)."""
    from PIL import Image

    if not isinstance(url, str) or not url.startswith("data:") or ";base64," not in url:
        raise BadRequest("only base64 data URLs are accepted for image_url (no network access from the server)")
    try:
        raw = base64.b64decode(url.split(";base64,", 1)[1], validate=False)
        Image.MAX_IMAGE_PIXELS = None
        return Image.open(io.BytesIO(raw)).convert("RGB")
    except Exception as ex:
        raise BadRequest(f"cannot decode image: {ex}") from ex


def build_prompt(messages):
    """Qwen2.5-VL chat template over OpenAI messages -> (prompt string, [PIL images in prompt order])."""
    if not isinstance(messages, list) or not messages:
        raise BadRequest("messages must be a non-empty list")
    parts, images = [], []
    if messages[0].get("role") != "system":
        parts.append(f"<|im_start|>system\n{DEFAULT_SYSTEM}<|im_end|>\n")
    for m in messages:
        role = m.get("role")
        if role not in ("system", "user", "assistant"):
            raise BadRequest(f"unsupported role {role!r}")
        content = m.get("content")
        body = []
        if isinstance(content, str):
            body.append(content)
        elif isinstance(content, list):
            for item in content:
                kind = item.get("type") if isinstance(item, dict) else None
                if kind == "text":
                    body.append(str(item.get("text", "")))
                elif kind == "image_url":
                    iu = item.get("image_url")
                    images.append(decode_data_url(iu.get("url") if isinstance(iu, dict) else iu))
                    body.append(IMAGE_PLACEHOLDER)
                else:
                    raise BadRequest(f"unsupported content item type {kind!r}")
        else:
            raise BadRequest("message content must be a string or a list of items")
        parts.append(f"<|im_start|>{role}\n{''.join(body)}<|im_end|>\n")
    parts.append("<|im_start|>assistant\n")
    return "".join(parts), images


class _Parsed:
    __slots__ = ("req", "prompt", "pil_images", "max_tokens", "sample", "temperature", "seed", "future")


class ChatServer:
    """Holds the model / processor pair; turns OpenAI request dicts into response dicts, one at a time
    (`complete`), as an explicit batch (`complete_many`) or through the batching dispatcher (`submit`)."""

    def __init__(self, model, processor, model_id: str = "ZoomEarth", batch_window_s: float = 0.01, max_batch=None):
        self.model, self.processor, self.model_id = model, processor, model_id
        self._lock = threading.Lock()  # the engine is single-stream
        self.batch_window_s = batch_window_s
        self.max_batch = int(max_batch or getattr(model.engine, "max_seqs", 1))
        self._queue = []
        self._cv = threading.Condition()
        self._worker = None
        self._stop = False
        self.scheduler = None  # the dispatcher's ChainScheduler (stats for tests / monitoring)

    # ------------------------------------------------------------------ request -> response pieces
    def _parse(self, req: dict) -> _Parsed:
        if req.get("stream"):
            raise BadRequest("stream=true is not offered")
        if int(req.get("n", 1) or 1) != 1:
            raise BadRequest("n must be 1")
        p = _Parsed()
        p.req = req
        p.prompt, p.pil_images = build_prompt(req.get("messages"))
        p.max_tokens = int(req.get("max_tokens") or req.get("max_completion_tokens") or 1024)
        t = req.get("temperature")
        p.sample = t is not None and float(t) > 0.0
        p.temperature = float(t) if p.sample else None
        p.seed = int(req.get("seed") or 0)
        p.future = None
        return p

    def _response(self, p: _Parsed, out, n_in: int) -> dict:
        eos = set(self.model.config.eos_token_ids)
        pad = self.model.config.pad_token_id
        out = out[: p.max_tokens]
        stop = next((i for i, t in enumerate(out) if t in eos), None)
        ids = out if stop is None else out[: stop + 1]
        while stop is None and ids and ids[-1] == pad:
            ids = ids[:-1]
        text = self.processor.tokenizer.decode(ids, skip_special_tokens=True).strip()
        return {
            "id": "chatcmpl-" + uuid.uuid4().hex[:24], "object": "chat.completion", "created": int(time.time()),
            "model": p.req.get("model") or self.model_id,
            "choices": [{"index": 0, "message": {"role": "assistant", "content": text},
                         "finish_reason": "stop" if stop is not None else "length"}],
            "usage": {"prompt_tokens": n_in, "completion_tokens": len(ids), "total_tokens": n_in + len(ids)},
        }

    def _run(self, batch):
        """One processor + generate call for the parsed requests of `batch` (all greedy, or a single request)."""
        from .image import DeviceImage

        with self._lock:
            images = [[DeviceImage.from_pil(im, self.model.engine) for im in p.pil_images] for p in batch]
            flat = [im for row in images for im in row]  # consumed in prompt order, row after row
            inputs = self.processor(text=[p.prompt for p in batch], images=flat or None, return_tensors="pt",
                                    padding="longest").to(self.model.device)
            width = int(inputs["input_ids"].shape[1])
            n_in = inputs["attention_mask"].sum(dim=1).tolist()
            p0 = batch[0]
            kw = dict(max_new_tokens=max(p.max_tokens for p in batch), num_beams=1, do_sample=p0.sample)
            if p0.sample:
                kw.update(temperature=p0.temperature, top_k=None, top_p=None, seed=p0.seed)
            out = self.model.generate(**inputs, **kw)[:, width:].tolist()
        return [self._response(p, row, int(n)) for p, row, n in zip(batch, out, n_in)]

    # ------------------------------------------------------------------ entry points
    def complete(self, req: dict) -> dict:
        return self._run([self._parse(req)])[0]

    def complete_many(self, reqs) -> list:
        """The requests as ONE batch (all must be greedy; at most max_seqs of them)."""
        batch = [self._parse(r) for r in reqs]
        if any(p.sample for p in batch) and len(batch) > 1:
            raise BadRequest("sampled requests are not batched")
        if len(batch) > self.max_batch:
            raise BadRequest(f"batch of {len(batch)} exceeds max_seqs = {self.max_batch}")
        return self._run(batch)

    def submit(self, req: dict):
        """Queues the request for the batching dispatcher; returns a concurrent.futures.Future of the response.
        Malformed requests raise BadRequest here, before anything is queued."""
        from concurrent.futures import Future

        p = self._parse(req)
        p.future = Future()
        with self._cv:
            if self._worker is None:
                self._worker = threading.Thread(target=self._dispatch, name="ze-batcher", daemon=True)
                self._worker.start()
            self._queue.append(p)
            self._cv.notify()
        return p.future

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify()

    def _dispatch(self):
        """Running-batch admission (the concurrency model of /root/reference/src/eval/infer_vllm.py:244-271, where the
        client keeps up to 100 requests in flight): greedy requests go to a `ChainScheduler` -- a request that arrives
        while others are decoding is prefilled and joins their next burst, one that finishes frees its KV slot at once --
        and each future resolves as soon as ITS chain ends.  Sampled requests (temperature / seed are baked into the captured
        decode step) wait for the running chains to drain; those that share a temperature and a seed then run together
        as a batch of their own."""
        from .image import DeviceImage
        from .scheduler import ChainScheduler, Request

        sched = None
        sampled, held = [], []
        while True:
            with self._cv:
                while not self._queue and not self._stop and not sampled and not held and not (sched is not None and sched.busy()):
                    self._cv.wait()
                if self._stop and not self._queue and not sampled and not held and not (sched is not None and sched.busy()):
                    return
                if sched is None or not sched.busy():  # idle: give concurrent arrivals a moment to share the first prefill
                    deadline = time.monotonic() + self.batch_window_s
                    while len(self._queue) < self.max_batch and not self._stop:
                        left = deadline - time.monotonic()
                        if left <= 0:
                            break
                        self._cv.wait(left)
                new, self._queue = self._queue, []
            with self._lock:
                if sched is None:
                    sched = ChainScheduler(self.model, self.processor, do_sample=False, max_batch=self.max_batch, burst=8)
                    self.scheduler = sched
                held.extend(new)
                new, held = (held, []) if not sampled else ([p for p in held if p.sample], [p for p in held if not p.sample])
                for p in new:  # (greedy arrivals wait behind a pending sampled request: it needs the engine alone)
                    if p.sample:
                        sampled.append(p)
                        continue

                    def done(req, tokens, text, p=p):
                        p.future.set_result(self._response(p, tokens, req.n_prompt))
                        return None

                    def failed(req, ex, p=p):
                        if not p.future.done():
                            p.future.set_exception(ex)

                    try:
                        imgs = [DeviceImage.from_pil(im, self.model.engine) for im in p.pil_images]
                        budget = max(1, min(p.max_tokens, self.model.engine.max_ctx))  # per request: never fails its batch
                        sched.submit(Request(prompt=p.prompt, images=imgs, max_new_tokens=budget, on_done=done,
                                             on_error=failed))
                    except Exception as ex:
                        failed(None, ex)
                if sched.busy():
                    try:
                        sched.step()
                    except Exception as ex:  # an engine failure: every request of the running batch gets the error
                        for r in sched.pending_requests():
                            if r.on_error:
                                r.on_error(r, ex)
                        sched = None
                    continue
            if sampled:
                # The running (greedy) batch has drained.  Temperature and seed are baked into the captured decode step, so
                # the sampled requests that share both run TOGETHER on a scheduler of their own (continuous batching, a
                # request's random stream = stream 0 of its seed: what it would draw running alone, so a request's tokens
                # do not depend on its company); other sampling configurations follow in turn.
                key = (sampled[0].temperature, sampled[0].seed)
                group = [p for p in sampled if (p.temperature, p.seed) == key]
                sampled = [p for p in sampled if (p.temperature, p.seed) != key]
                with self._lock:
                    ss = ChainScheduler(self.model, self.processor, do_sample=True, temperature=key[0], seed=key[1],
                                        max_batch=self.max_batch, burst=8)
                    for p in group:
                        def done(req, tokens, text, p=p):
                            p.future.set_result(self._response(p, tokens, req.n_prompt))
                            return None

                        def failed(req, ex, p=p):
                            if not p.future.done():
                                p.future.set_exception(ex)

                        try:
                            imgs = [DeviceImage.from_pil(im, self.model.engine) for im in p.pil_images]
                            ss.submit(Request(prompt=p.prompt, images=imgs, max_new_tokens=max(1, min(p.max_tokens, self.model.engine.max_ctx)),
                                              stream_id=0, on_done=done, on_error=failed))
                        except Exception as ex:
                            failed(None, ex)
                    try:
                        ss.run()
                    except Exception as ex:
                        for r in ss.pending_requests():
                            if r.on_error:
                                r.on_error(r, ex)
                    for p in group:  # (belt and braces: nobody is left waiting)
                        if not p.future.done():
                            p.future.set_exception(RuntimeError("request was dropped by the sampled batch"))
                    sched = None  # the sampled scheduler used the chain slots: the greedy one starts afresh


def create_app(server: ChatServer):
    from fastapi import FastAPI, Request
    from fastapi.responses import JSONResponse

    app = FastAPI(title="zoomearth-mi355x")

    @app.get("/health")
    def health():
        return {"status": "ok"}

    @app.get("/v1/models")
    def models():
        return {"object": "list", "data": [{"id": server.model_id, "object": "model", "owned_by": "zoomearth-mi355x"}]}

    @app.post("/v1/chat/completions")
    async def chat(request: Request):
        import asyncio

        try:
            body = await request.json()
            return JSONResponse(await asyncio.wrap_future(server.submit(body)))
        except BadRequest as ex:
            return JSONResponse({"error": {"message": str(ex), "type": "invalid_request_error"}}, status_code=400)
        except Exception as ex:  # engine errors surface as a 500 with the engine's message
            return JSONResponse({"error": {"message": str(ex), "type": "server_error"}}, status_code=500)

    return app


def main():  # pragma: no cover
    import argparse

    import uvicorn

    from .modeling import ZoomEarthForConditionalGeneration
    from .processor import ZoomEarthProcessor

    ap = argparse.ArgumentParser(description="OpenAI-compatible server on the MI355X engine")
    ap.add_argument("--model_name", required=True)
    ap.add_argument("--served_model_name", default="ZoomEarth")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    args = ap.parse_args()
    model = ZoomEarthForConditionalGeneration.from_pretrained(args.model_name)
    processor = ZoomEarthProcessor.from_pretrained(args.model_name, trust_remote_code=True, max_pixels=128 * 128 * 28 * 28)
    processor.tokenizer.padding_side = "left"
    uvicorn.run(create_app(ChatServer(model, processor, args.served_model_name)), host=args.host, port=args.port)


if __name__ == "__main__":  # pragma: no cover
    main()
