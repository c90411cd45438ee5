# generic image targets (reference include/build.mk): make build image=<name> dockerpath=<ctx>
build:
	docker build -t $(image) -f $(dockerfile) $(dockerpath)

push:
	docker push $(image)

.PHONY: build push
