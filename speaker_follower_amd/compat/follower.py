"""`from follower import ...` (train.py:16, rational_follower.py:8, data_augmentation_from_speaker.py:5)."""
from speaker_follower_amd.agents import BaseAgent, Seq2SeqAgent, path_element_from_observation      # noqa: F401
from speaker_follower_amd.follower import batch_instructions_from_encoded                           # noqa: F401
from speaker_follower_amd.search import (least_common_viewpoint_path, backchain_inference_states,   # noqa: F401
                                         InferenceState)
