"""Host runtime of the MI355X path: model dims, weight preparation, the ctypes binding of
`librs_asr.so` and the batched pipeline that drives it.  PyTorch-ROCm is used for device
memory, streams and `torch.distributed` only — every FLOP runs in the HIP library."""
