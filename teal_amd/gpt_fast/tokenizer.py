"""Minimal tokenizer wrapper for real checkpoints (sentencepiece `tokenizer.model`).  Out of the hot
path's scope; synthetic runs use raw token ids.  Role of gpt-fast/tokenizer.py:get_tokenizer."""
from pathlib import Path


class SentencePieceWrapper:
    def __init__(self, model_path: Path):
        from sentencepiece import SentencePieceProcessor
        self.processor = SentencePieceProcessor(str(model_path))

    def encode(self, text: str):
        return self.processor.EncodeAsIds(text)

    def decode(self, tokens):
        return self.processor.DecodeIds(tokens)

    def bos_id(self):
        return self.processor.bos_id()

    def eos_id(self):
        return self.processor.eos_id()


def get_tokenizer(tokenizer_model_path, model_name):
    if "llama-3" in str(model_name).lower():
        raise NotImplementedError("tiktoken (Llama-3) tokenizers are not available in this image; pass token ids")
    return SentencePieceWrapper(Path(tokenizer_model_path))
