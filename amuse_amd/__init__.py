"""amuse_amd - AMUSE's latent-diffusion gesture sampling on MI355X: the host side above the C ABI of libamuse_hip.so (include/amuse_hip.h).

Nothing is imported here: `amuse_amd._lib` loads the library on first use and fails loudly when it is missing; the modules mirror the reference's interface
(infer_ldm, ldm, main, trainer, train_gesture, ...)."""
