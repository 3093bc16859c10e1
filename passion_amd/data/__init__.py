"""training input pipeline on the GPU: passion_amd.data.transforms"""
