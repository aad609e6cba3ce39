"""semi_tts_amd -- MI355X-native (gfx950) implementation of the Tacotron-style TTS decode hot
path and the VQ codebook lookup of ttaoREtw/semi-tts, behind the reference's module API.

    from semi_tts_amd.tts import Tacotron2
    from semi_tts_amd.embed import L2Embedding, SeperateEmbedding

Compute goes through libsemitts_hip.so (hand-written HIP, C ABI in include/semitts.h).
"""
__all__ = ['tts', 'module', 'embed', 'ops', 'synthetic', 'build']
