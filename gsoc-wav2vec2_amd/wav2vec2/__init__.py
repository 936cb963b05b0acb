"""MI355X-native drop-in for the reference's ``wav2vec2`` package surface
(src/wav2vec2/__init__.py:1-4): the same six names."""

from .config import RobustWav2Vec2Config, Wav2Vec2Config
from .losses import CTCLoss
from .modeling import Wav2Vec2ForCTC, Wav2Vec2Model
from .processor import Wav2Vec2Processor
from .training import Trainer

__all__ = ["Wav2Vec2Config", "RobustWav2Vec2Config", "CTCLoss", "Wav2Vec2ForCTC", "Wav2Vec2Model",
           "Wav2Vec2Processor", "Trainer"]
