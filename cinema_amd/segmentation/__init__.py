"""Segmentation model of the CineMA family on the HIP tape (reference ``cinema/segmentation``)."""
