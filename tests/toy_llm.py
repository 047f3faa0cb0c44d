"""A random-init tiny LlamaForCausalLM (transformers is importable; nothing is downloaded) and a toy word tokenizer with the
tokenizer surface MiniGPTBase uses (`__call__(text, return_tensors="pt", add_special_tokens=...)`, `.input_ids`, `.to`,
`decode(ids, skip_special_tokens=True)`): stand-ins for Vicuna-7B + LlamaTokenizer, which are not in the container."""
import torch

VOCAB = 96
PAD, BOS, EOS = 0, 1, 2


from certifiedgpt_amd.minigpt4 import WordHashTokenizer


class ToyTokenizer(WordHashTokenizer):
    def __init__(self):
        super().__init__(VOCAB)


def tiny_llama(hidden=64, seed=0, dtype=torch.float32, device="cpu"):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(vocab_size=VOCAB, hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=2,
                      num_attention_heads=4, num_key_value_heads=4, max_position_embeddings=512, pad_token_id=PAD,
                      bos_token_id=BOS, eos_token_id=EOS)
    torch.manual_seed(seed)
    return LlamaForCausalLM(cfg).to(device=device, dtype=dtype).eval()


class StubEncoder:
    """encode_img stand-in for CPU tests: a fixed random projection of the mean-pooled image to [B, queries, hidden]."""
    max_batch = 8

    def __init__(self, hidden=64, queries=4, seed=1):
        g = torch.Generator().manual_seed(seed)
        self.w = torch.randn(3, queries * hidden, generator=g)
        self.q, self.h = queries, hidden

    def encode_img(self, images):
        emb = (images.mean(dim=(2, 3)) @ self.w).view(-1, self.q, self.h)
        return emb, torch.ones(emb.shape[:-1], dtype=torch.long)
