"""The column shard of the N > 1 path with its three library-facing methods pointed at the CPU oracle and CPU tensors, so that
world-size-2 gloo tests can run the product class (`dsa_amd.sharding.ColumnShard`: ranges, slices, the three reduction schedules,
write routing) in a container without a GPU.  Test infrastructure."""
import numpy as np


def make_cpu_shard_class(sharding):
    class CpuColumnShard(sharding.ColumnShard):
        def _build(self, I, J, V, local_columns):
            if local_columns:
                return self.api.dynamicsparse(I, J, V, self.m, self.ncols, binding=self.binding)
            mine = (J > self.col0) & (J <= self.col0 + self.ncols)
            # local column keys 1..ncols: the shard is the reference layout of its own sub-matrix
            return self.api.dynamicsparse(I[mine], J[mine] - self.col0, V[mine], self.m, self.ncols, binding=self.binding)

        def _device_of(self, device):
            import torch
            return torch.device("cpu")

        def spmv_partial(self, x_local, y=None):
            import torch
            if y is None:
                y = self.new_y()
            y.copy_(torch.from_numpy(self.A.mul(x_local.numpy(), dense_out=self.m)))
            return y

    return CpuColumnShard
