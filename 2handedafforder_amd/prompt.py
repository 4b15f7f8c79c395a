"""Host-side prompt helpers mirroring the reference's (tiny, pure Python, off the GPU path).

  tokenizer_image_token  <- 2Haff/model/llava/mm_utils.py:19-44
  conv_llava_v1          <- 2Haff/model/llava/conversation.py:355-365 (two-separator style, get_prompt :31-105)
  build_*_prompt         <- inference.py:221-226 (no template) and chat.py:155-168 (llava_v1 template)
"""
from dataclasses import dataclass, field
from typing import List

IMAGE_TOKEN_INDEX = -200          # utils/utils.py:8
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """Tokenise the text around every "<image>" and put `image_token_index` in its place, keeping one BOS."""
    pieces = [tokenizer(chunk).input_ids for chunk in prompt.split(DEFAULT_IMAGE_TOKEN)]
    has_bos = bool(pieces) and bool(pieces[0]) and pieces[0][0] == tokenizer.bos_token_id
    skip = 1 if has_bos else 0
    ids = [pieces[0][0]] if has_bos else []
    for n, piece in enumerate(pieces):
        if n > 0:
            ids.append(image_token_index)
        ids.extend(piece[skip:])
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        import torch
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


@dataclass
class TwoSepConversation:
    system: str
    roles: tuple
    sep: str
    sep2: str
    messages: List[list] = field(default_factory=list)

    def append_message(self, role, message):
        self.messages.append([role, message])

    def get_prompt(self):
        out = self.system + self.sep
        for i, (role, message) in enumerate(self.messages):
            out += f"{role}: {message}{(self.sep, self.sep2)[i % 2]}" if message else f"{role}:"
        return out

    def copy(self):
        return TwoSepConversation(self.system, self.roles, self.sep, self.sep2, [list(m) for m in self.messages])


def conv_llava_v1():
    return TwoSepConversation(
        system="A chat between a curious human and an artificial intelligence assistant. "
               "The assistant gives helpful, detailed, and polite answers to the human's questions.",
        roles=("USER", "ASSISTANT"), sep=" ", sep2="</s>")


def image_placeholder(use_mm_start_end=True):
    return (DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN) if use_mm_start_end else DEFAULT_IMAGE_TOKEN


def build_inference_prompt(narration, use_mm_start_end=True):
    """inference.py:221-226 — no conversation template."""
    return image_placeholder(use_mm_start_end) + "\nWhere would you interact with the object to perform action " + narration


def build_chat_prompt(user_text, use_mm_start_end=True):
    """chat.py:155-168 — llava_v1 template around '<image>\\n' + text."""
    conv = conv_llava_v1()
    conv.append_message(conv.roles[0], image_placeholder(use_mm_start_end) + "\n" + user_text)
    conv.append_message(conv.roles[1], "")
    return conv.get_prompt()
