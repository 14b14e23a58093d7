"""CPU oracle for the Spherical-DYffusion sampling path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`spherical-dyffusion_amd/`) imports this package; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and
there only as the checker / reported baseline -- never as the thing measured
or shipped.

The oracle is a plain PyTorch-CPU fp32 restatement (the reference itself is
PyTorch, so this is the same arithmetic: `torch.fft` + `einsum` + `conv2d` +
`instance_norm` + exact-erf GELU) of

  * `torch_harmonics.RealSHT / InverseRealSHT`  (third-party, un-vendored and
    un-pinned in the reference: `setup.py:98`, call sites
    `src/models/sfno/sfnonet.py:551-554`)                -> `oracle/sht.py`
  * `SpectralConvS2`, `FourierNeuralOperatorBlock`,
    `SphericalFourierNeuralOperatorNet`
    (`src/models/sfno/s2convolutions.py:158-193`,
    `src/models/sfno/sfnonet.py:289-337,797-841`)        -> `oracle/sfno.py`
  * `BaseDYffusion.sample_loop` / `DYffusion._interpolate`
    (`src/diffusion/dyffusion.py:457-567,642-662`)       -> `oracle/dyffusion.py`
  * the counter-based dropout stream of the product      -> `oracle/philox.py`

Pinning status (see DESIGN.md "Oracle"):
  * network + sampler: pinned against golden vectors produced by importing the
    reference's own classes in the build container
    (`tools/gen_golden.py` -> `tests/golden/*.npz`).
  * SHT: "parity unpinned" by the reference (the package is absent from
    /root/reference and from this image); pinned analytically instead
    (orthonormality, scipy `sph_harm` known answer, exact quadrature moments).
"""
