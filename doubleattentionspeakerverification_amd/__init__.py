"""MI355X-native hot path of fedecosta/DoubleAttentionSpeakerVerification.

Public surface = the reference's module API for the path BASELINE.json names:
``poolings`` (DoubleMHA & friends, hand-written HIP), ``CNNs`` (VGG front-ends),
``loss`` (AM-softmax), ``model`` (SpeakerClassifier), plus ``dist`` (one-process-per-GPU
data parallelism over RCCL).  The compute library is libdasv_hip.so (include/dasv_hip.h).
"""
__version__ = "0.1.0"
