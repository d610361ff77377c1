// archs.h -- dense stacks that get a fully fused forward kernel (fused_fwd.h).
// Any other (dims, act) list runs through the generic per-layer path (gemm.h).
//   S1  BASELINE.json configs[1]: 7 -> [352,352,352,224] -> 451
//   S2  reference default DirectEmulator (emulator.py:196): 7 -> [288,352,288,224] -> 451
//   S3  AutoEncoderEmulator.predict chain (emulator.py:789-790): latent emulator
//       7 -> [352,352,352,224] -> 9 (linear) followed by decoder 9 -> [32,352] -> 451
//   S4  decoder alone (emulator.py:522-524): 9 -> [32,352] -> 451
#pragma once
namespace v21 {
struct ArchS1 { static constexpr int L = 5; static constexpr int dims[6] = {7, 352, 352, 352, 224, 451}; static constexpr int act[5] = {1, 1, 1, 1, 0}; };
struct ArchS2 { static constexpr int L = 5; static constexpr int dims[6] = {7, 288, 352, 288, 224, 451}; static constexpr int act[5] = {1, 1, 1, 1, 0}; };
struct ArchS3 { static constexpr int L = 8; static constexpr int dims[9] = {7, 352, 352, 352, 224, 9, 32, 352, 451}; static constexpr int act[8] = {1, 1, 1, 1, 0, 1, 1, 0}; };
struct ArchS4 { static constexpr int L = 3; static constexpr int dims[4] = {9, 32, 352, 451}; static constexpr int act[3] = {1, 1, 0}; };
}  // namespace v21
#define V21_ARCH_LIST(X) X(S1) X(S2) X(S3) X(S4)
// stacks with a compiled fused TRAINING kernel (fused_train.h: large steps of f16 / bf16 trainers):
//   T1  the autoencoder 451 -> [352] -> 9 -> [32, 352] -> 451 (emulator.py:522-524; BASELINE configs[3])
//   T2  the latent emulator 7 -> [352,352,352,224] -> 9 (emulator.py:525)
//   T3  the reference's default direct emulator 7 -> [288,352,288,224] -> 451 (emulator.py:196)
//   T4  BASELINE configs[1]'s direct emulator 7 -> [352,352,352,224] -> 451
namespace v21 {
struct ArchT1 { static constexpr int L = 5; static constexpr int dims[6] = {451, 352, 9, 32, 352, 451}; static constexpr int act[5] = {1, 0, 1, 1, 0}; };
struct ArchT2 { static constexpr int L = 5; static constexpr int dims[6] = {7, 352, 352, 352, 224, 9}; static constexpr int act[5] = {1, 1, 1, 1, 0}; };
struct ArchT3 { static constexpr int L = 5; static constexpr int dims[6] = {7, 288, 352, 288, 224, 451}; static constexpr int act[5] = {1, 1, 1, 1, 0}; };
struct ArchT4 { static constexpr int L = 5; static constexpr int dims[6] = {7, 352, 352, 352, 224, 451}; static constexpr int act[5] = {1, 1, 1, 1, 0}; };
}  // namespace v21
#define V21_TRAIN_ARCH_LIST(X) X(T1) X(T2) X(T3) X(T4)
