// ShaderConstants.h — POD mirrors of the per-shader constant buffers (CONSTANT_BUFFER_SHADER, b0)
// that the reference's pass classes memcpy to the GPU.  Layout = the C++ structs in
// Engine/Include/Renderer/Pipeline/DeferredPipeline.h (cited per struct).
#pragma once
#include <cstdint>

namespace MRendererHip {

struct Vector2 { float x, y; };

struct PreFilterEnvMapConstant { float Roughness; uint32_t MipLevel; uint32_t EnvMapSize; };       // DeferredPipeline.h:46-51
struct PrecomputeBRDFConstant { uint32_t TextureResolution; };                                        // :75-78
struct BloomPrefilterConstant { Vector2 TexelSize; float Threshold; float Knee; };                    // :216-221
struct BlurConstant { Vector2 TexelSize; };                                                           // :232-235, :259-262
struct ClusteredShaderConstant { int32_t NumLight; };                                                 // :303-306
struct LuminanceHistogramConstant { uint32_t TextureWidth, TextureHeight; float MinLogLuminance, InvLogLuminanceRange; };   // :377-383
struct AverageLuminanceConstant { uint32_t PixelCount; float MinLogLuminance, LogLuminanceRange; };   // :392-397

}  // namespace MRendererHip
