"""`from model import ...` (train.py:15, train_speaker.py:15) -> speaker_follower_amd.model."""
from speaker_follower_amd.model import (EncoderLSTM, AttnDecoderLSTM, SpeakerEncoderLSTM,     # noqa: F401
                                        SpeakerDecoderLSTM, SoftDotAttention, VisualSoftDotAttention,
                                        EltwiseProdScoring)
