"""`from speaker import Seq2SeqSpeaker` (train_speaker.py:16, data_augmentation_from_speaker.py:4)."""
from speaker_follower_amd.agents import Seq2SeqSpeaker                                              # noqa: F401
from speaker_follower_amd.follower import batch_instructions_from_encoded                           # noqa: F401
