// Model container (.jupw) reader, BatchNorm folding and kernel-ready weight
// packing.  The container is this engine's replacement for the serialized
// TensorRT engine the reference loads (reference core/src/core.cc:156-167,
// core/src/tensorrt_backend.cc:117-148); its Python twin is
// joshupscale_amd/model_file.py.
#pragma once

#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"

namespace ju {

struct ModelConfig {
	int frameHeight = 0;
	int frameWidth = 0;
	int numFlowInputs = 4;
	int flowArch = 0;  // 0 = autoencoder, 1 = resnet
	int flowPadFactor = 0;
	bool normalizeBrightness = false;
	int genFilters = 64;
	int genBlocks = 24;
	int flowResFilters = 64;
	int flowResBlocks = 10;
	std::vector<int> flowFilters;
	float bnEps = 1e-3f;
	int computeDtype = kBF16;
	// temporal moving-average output filter (frame_moving_avg.py); strength 0 = off
	float temporalStrength = 0.0f;
	float temporalThreshold = 0.1f;
	// its other switches (frame_moving_avg.py:99-110): --window (HR pixels, 0 = global gate),
	// --gain (0 = sign gate), --norm L2, --limit, --luma-normalize
	int temporalWindow = 0;
	float temporalGain = 0.0f;
	bool temporalL2 = false, temporalLimit = false, temporalLuma = false;
	// `activation` of the sub-model constructors (reference models.py:24-27, 261, 337, 489):
	// 0 = relu, 1 = lrelu (keras LeakyReLU(negative_slope))
	int flowActivation = 0, genActivation = 0;
	float flowNegativeSlope = 0.0f, genNegativeSlope = 0.0f;

	// reference scripts/training/models.py:735-744
	int paddedHeight() const {
		return flowPadFactor ? (frameHeight + flowPadFactor - 1) / flowPadFactor * flowPadFactor
		                     : frameHeight;
	}
	int paddedWidth() const {
		return flowPadFactor ? (frameWidth + flowPadFactor - 1) / flowPadFactor * flowPadFactor
		                     : frameWidth;
	}
};

struct TensorView {
	std::vector<int> dims;
	const float *data = nullptr;
	std::size_t count = 0;
};

class ModelFile {
public:
	// Parses and validates; keeps its own copy of the bytes.
	ModelFile(const void *blob, std::size_t size);

	const ModelConfig &config() const { return m_Config; }
	bool has(const std::string &name) const { return m_Tensors.count(name) != 0; }
	const TensorView &tensor(const std::string &name) const;  // throws if missing
	// Tensor that must have exactly these dims.
	const TensorView &tensor(const std::string &name, const std::vector<int> &dims) const;

private:
	std::vector<unsigned char> m_Bytes;
	ModelConfig m_Config;
	std::map<std::string, TensorView> m_Tensors;
};

// A convolution with BatchNorm folded in: y = sum_k x[k] * w[tap][k][cout] + bias.
struct FoldedConv {
	int taps = 9;
	int cin = 0;
	int cout = 0;
	std::vector<float> w;     // [taps][cin][cout]
	std::vector<float> bias;  // [cout]
};

// Conv2D (keras kernel [k][k][cin][cout]) followed by an optional BatchNorm
// (folded: W' = W*g/sqrt(var+eps), b' = beta - mean*g/sqrt(var+eps))
// and/or the layer's own bias.  `bnPrefix` empty = no BN.
FoldedConv foldConv(const ModelFile &m, const std::string &convName,
    const std::string &bnPrefix, bool hasBias);

// Conv2DTranspose k2 s2 (keras kernel [2][2][cout][cin]) + BN, expressed as a
// 1x1 convolution cin -> 4*cout with output channel (a*2+b)*cout + o.
FoldedConv foldConvTranspose2x2(
    const ModelFile &m, const std::string &convName, const std::string &bnPrefix);

// Every convolution of the model, folded and checked, in execution order.  Host
// only (no device): this is what ju_validate_model runs and what the engine uploads.
struct ConvSpec {
	std::string name;         // container layer name, e.g. "generator/block_3/conv_1"
	FoldedConv conv;
	std::vector<int> cinMap;  // packed input channel -> source channel (-1 = zero), see packConvWeights
	int H = 0, W = 0;         // resolution the layer runs at
};

// Range checks on the header fields (sizes, filter counts, divisibility of the padded
// frame by the auto-encoder depth, ...).  Throws std::invalid_argument.
void validateConfig(const ModelConfig &c);

// Folds every layer and checks the whole channel chain (each layer's cin is the
// previous layer's cout, kernel sizes, head widths).  Throws std::invalid_argument
// naming the offending layer.  Also returns the raw convT2 kernel/bias views through
// the ModelFile (generator/conv_trans_2/{kernel,bias}, shapes checked).
std::vector<ConvSpec> foldModel(const ModelFile &m);

// Kernel-ready 16-bit weights for conv_mfma_kernel:
// [cout/COG][cinP/CK][tap][CK/16][2][COG][8] with COG = 32*nb (the cout block the
// launch will use, see convTiling), CK = convCK(cinP).  `cinMap[k]` = source input
// channel of packed channel k, or -1 for a zero channel (cinMap.size() == cinP,
// a multiple of 16).
std::vector<std::uint16_t> packConvWeights(
    const FoldedConv &c, const std::vector<int> &cinMap, int nb, DType dt);

// A 3x3 convolution of 64 (padded) input and 64 output channels as the resident tower's A fragments in the
// v_mfma_f32_16x16x32 shape (tower_kernels.hip, JU_TOWER_M16): [tap 9][ks32 2][c16 2][ch 2][lane 64][8] --
// lane l of fragment (tap, ks32, c16) of wave half ch holds output channel ch * 32 + c16 * 16 + (l & 15), packed input
// channels ks32 * 32 + 8 (l >> 4) .. + 7.  Same element count and per-layer size as packConvWeights(nb = 2).
std::vector<std::uint16_t> packTowerWeightsM16(const FoldedConv &c, const std::vector<int> &cinMap, DType dt);

// convT2 (keras kernel [2][2][3][32]) as MFMA A fragments for tail_fused_kernel:
// A[m][k], m = 4*(a'*2+b') + c (c < 3, other rows zero), k = input channel;
// [2 k-steps][64 lanes][8]: lane l holds A[l & 31][16 ks + 8 (l >> 5) + j].
std::vector<std::uint16_t> packTailWeights(const float *k2, DType dt);

std::uint16_t floatToF16(float f);
std::uint16_t floatToBF16(float f);

}  // namespace ju
