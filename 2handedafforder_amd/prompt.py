"""Host-side prompt helpers mirroring the reference's (tiny, pure Python, off the GPU path).

  tokenizer_image_token  <- 2Haff/model/llava/mm_utils.py:19-44
  conv_llava_v1          <- 2Haff/model/llava/conversation.py:355-365 (two-separator style, get_prompt :31-105)
  conv_llava_llama_2     <- conversation.py:300-311 (LLAMA_2 style, get_prompt :72-93)
  conv_templates         <- the two keys the CLIs' --conv_type accepts (chat.py:41-46, train_ds.py:115-120) of conversation.py:381-395
  label_separator        <- utils/dataset.py:97-101 (what splits a round into instruction | answer for the label mask)
  build_*_prompt         <- inference.py:221-226 (no template) and chat.py:155-168 (the --conv_type template)
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


@dataclass
class Llama2Conversation:
    """SeparatorStyle.LLAMA_2: `<s>[INST] <<SYS>>\nsystem\n<</SYS>>\n\nuser [/INST] answer </s><s>[INST] user [/INST] ...` with
    the leading `<s>` stripped (the tokenizer adds BOS); an empty message adds nothing."""
    system: str
    roles: tuple
    sep: str
    sep2: str
    messages: List[list] = field(default_factory=list)

    def append_message(self, role, message):
        self.messages.append([role, message])

    def get_prompt(self):
        out = ""
        for i, (role, message) in enumerate(self.messages):
            if i == 0:
                assert message, "first message should not be none"
                assert role == self.roles[0], "first message should come from user"
            if not message:
                continue
            if i == 0:
                message = f"<<SYS>>\n{self.system}\n<</SYS>>\n\n" + message
            out += (self.sep + f"[INST] {message} [/INST]") if i % 2 == 0 else (" " + message + " " + self.sep2)
        return out.lstrip(self.sep)      # str.lstrip with a character SET, as the reference does (conversation.py:93)

    def copy(self):
        return Llama2Conversation(self.system, self.roles, self.sep, self.sep2, [list(m) for m in self.messages])


def conv_llava_llama_2():
    return Llama2Conversation(
        system="You are a helpful language and vision assistant. "
               "You are able to understand the visual content that the user provides, "
               "and assist the user with a variety of tasks using natural language.",
        roles=("USER", "ASSISTANT"), sep="<s>", sep2="</s>")


def conv_llava_v1():
    return TwoSepConversation(
        system="A chat between a curious human and an artificial intelligence assistant. "
               "The assistant gives helpful, detailed, and polite answers to the human's questions.",
        roles=("USER", "ASSISTANT"), sep=" ", sep2="</s>")


conv_templates = {"llava_v1": conv_llava_v1, "llava_llama_2": conv_llava_llama_2}


def get_conv(conv_type="llava_v1"):
    """A fresh conversation of the template --conv_type names (conversation_lib.conv_templates[args.conv_type].copy(), chat.py:155,
    train_ds.py:188-190). The CLIs offer exactly these two (argparse choices); anything else is an error, never a silent llava_v1."""
    try:
        return conv_templates[conv_type]()
    except KeyError:
        raise ValueError(f"unknown conv_type {conv_type!r}: expected one of {sorted(conv_templates)}") from None


_default_conv_type = "llava_v1"


def set_default_conversation(conv_type):
    """train_ds.py:188-190: `conversation_lib.default_conversation = conv_templates[args.conv_type]` — what the datasets template
    their question / answer pairs with (utils/aff_dataset.py:251, `default_conversation.copy()`)."""
    global _default_conv_type
    get_conv(conv_type)      # validates
    _default_conv_type = conv_type


def default_conversation():
    return get_conv(_default_conv_type)


def label_separator(conv_type, conv):
    """utils/dataset.py:97-101: the text that ends the instruction part of a round."""
    return (conv.sep + conv.roles[1] + ": ") if conv_type == "llava_v1" else "[/INST] "


def image_placeholder(use_mm_start_end=True):
    return (DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN) if use_mm_start_end else DEFAULT_IMAGE_TOKEN


def build_inference_prompt(narration, use_mm_start_end=True):
    """inference.py:221-226 — no conversation template."""
    return image_placeholder(use_mm_start_end) + "\nWhere would you interact with the object to perform action " + narration


def build_chat_prompt(user_text, use_mm_start_end=True, conv_type="llava_v1"):
    """chat.py:155-168 — the --conv_type template around '<image>\\n' + text."""
    conv = get_conv(conv_type)
    conv.append_message(conv.roles[0], image_placeholder(use_mm_start_end) + "\n" + user_text)
    conv.append_message(conv.roles[1], "")
    return conv.get_prompt()
