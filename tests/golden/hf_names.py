"""Tensor-name mapping from the whisper.cpp / openai names used in this repo to HuggingFace state-dict names
(shared by the golden-vector generators)."""


def hf_name(n: str) -> str:
    n = n.replace("encoder.blocks.", "model.encoder.layers.").replace("decoder.blocks.", "model.decoder.layers.")
    n = n.replace(".cross_attn_ln.", ".encoder_attn_layer_norm.").replace(".attn_ln.", ".self_attn_layer_norm.")
    n = n.replace(".cross_attn.", ".encoder_attn.").replace(".attn.", ".self_attn.")
    n = n.replace(".query.", ".q_proj.").replace(".key.", ".k_proj.").replace(".value.", ".v_proj.").replace(".out.", ".out_proj.")
    n = n.replace(".mlp_ln.", ".final_layer_norm.").replace(".mlp.0.", ".fc1.").replace(".mlp.2.", ".fc2.")
    n = n.replace("encoder.conv", "model.encoder.conv").replace("encoder.positional_embedding", "model.encoder.embed_positions.weight")
    n = n.replace("encoder.ln_post.", "model.encoder.layer_norm.").replace("decoder.token_embedding.weight", "model.decoder.embed_tokens.weight")
    n = n.replace("decoder.positional_embedding", "model.decoder.embed_positions.weight").replace("decoder.ln.", "model.decoder.layer_norm.")
    return n
