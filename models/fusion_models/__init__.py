"""Drop-in module path of the reference (DanielMengLiu/DeepLip); implementations live in deeplip_amd."""
