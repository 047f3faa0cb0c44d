"""ORACLE (test infrastructure) -- restatement of `MiniGPTBase.generate` and `get_context_emb`
(reference graphs/models/minigpt4/models/minigpt_base.py:374-448 and :75-89) with `encode_img`'s output given.

Only tests may import this.  It follows the reference statement by statement (one context embedding per sample, left
padding into a zero tensor, Hugging Face `generate` with the reference's fixed arguments, the same decode clean-up); the
product (certifiedgpt_amd/minigpt4.py) embeds a shared prompt once and broadcasts it.  Pinned by:
tests/golden/generate_golden.{json,npz}, produced by oracle/gen_golden_generate.py, which EXECUTES the reference's own
`MiniGPTBase.generate` / `get_context_emb` (minigpt_base.py loaded by file path, unbound methods on a bare instance; CPU fp32, a
random-init tiny Llama + the toy tokenizer, shared / ragged / single prompts, prescribed token rows and decoded strings for the
clean-up lines): tests/test_minigpt4_cpu.py checks this restatement AND the product against it, answer by answer.
"""
import torch


def get_context_emb(embed_tokens, tokenizer, prompt, img_list):
    """minigpt_base.py:75-89."""
    device = img_list[0].device
    prompt_segs = prompt.split('<ImageHere>')
    assert len(prompt_segs) == len(img_list) + 1, "Unmatched numbers of image placeholders and images."
    seg_tokens = [
        tokenizer(seg, return_tensors="pt", add_special_tokens=i == 0).to(device).input_ids   # only add bos to the first seg
        for i, seg in enumerate(prompt_segs)
    ]
    seg_embs = [embed_tokens(seg_t) for seg_t in seg_tokens]
    mixed_embs = [emb for pair in zip(seg_embs[:-1], img_list) for emb in pair] + [seg_embs[-1]]
    return torch.cat(mixed_embs, dim=1)


@torch.no_grad()
def generate(llama_model, tokenizer, img_embeds, texts, num_beams=1, max_new_tokens=20, min_length=1, top_p=0.9,
             repetition_penalty=1, length_penalty=1, temperature=1, do_sample=False):
    """minigpt_base.py:374-448 from the line after `encode_img` (:401) on."""
    embed_tokens = llama_model.get_input_embeddings()
    image_lists = [[image_emb[None]] for image_emb in img_embeds]                               # :402
    batch_embs = [get_context_emb(embed_tokens, tokenizer, text, img_list) for text, img_list in zip(texts, image_lists)]
    batch_size = len(batch_embs)
    max_len = max([emb.shape[1] for emb in batch_embs])
    emb_dim = batch_embs[0].shape[2]
    dtype = batch_embs[0].dtype
    device = batch_embs[0].device
    embs = torch.zeros([batch_size, max_len, emb_dim], dtype=dtype, device=device)
    attn_mask = torch.zeros([batch_size, max_len], dtype=torch.int, device=device)
    for i, emb in enumerate(batch_embs):                                                        # :413-416
        emb_len = emb.shape[1]
        embs[i, -emb_len:] = emb[0]
        attn_mask[i, -emb_len:] = 1
    # :418-431.  The reference passes `min_length=min_length`; its pinned transformers (4.30.0, docker/tpu-docker:32) counts GENERATED
    # tokens for an inputs_embeds call, later versions subtract the embedded prompt's length (5.15: min_length 1 -> 0).  `min_new_tokens`
    # says "generated tokens" in every version, so the oracle states the reference's semantics independently of what is installed.
    outputs = llama_model.generate(inputs_embeds=embs, attention_mask=attn_mask, max_new_tokens=max_new_tokens,
                                   num_beams=num_beams, length_penalty=length_penalty, temperature=temperature,
                                   do_sample=do_sample, min_new_tokens=min_length, top_p=top_p,
                                   repetition_penalty=repetition_penalty)
    answers = []
    for output_token in outputs:                                                                # :441-448
        if output_token[0] == 0:
            output_token = output_token[1:]
        output_texts = tokenizer.decode(output_token, skip_special_tokens=True)
        output_texts = output_texts.split('</s>')[0]
        output_texts = output_texts.replace("<s>", "")
        output_texts = output_texts.split(r'[/INST]')[-1].strip()
        answers.append(output_texts)
    return answers
